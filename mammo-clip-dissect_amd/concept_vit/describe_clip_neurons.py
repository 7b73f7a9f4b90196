"""Drop-in for the reference's concept_vit/describe_clip_neurons.py (vanilla CLIP-Dissect driver): same flags
and defaults (:11-35), top-1 concept per neuron -> descriptions.csv + args.txt (:49-93)."""
import argparse

from . import CLIP_og_utils
from ._driver import describe_layers, setup_device, write_results

parser = argparse.ArgumentParser(description='CLIP-Dissect')
parser.add_argument("--clip_model", type=str, default="ViT-B/16",
                    choices=['RN50', 'RN101', 'RN50x4', 'RN50x16', 'RN50x64', 'ViT-B/32', 'ViT-B/16', 'ViT-L/14'],
                    help="Which CLIP-model to use")
parser.add_argument("--target_model", type=str, default="resnet50", help="Which model to dissect")
parser.add_argument("--target_layers", type=str, default="conv1,layer1,layer2,layer3,layer4",
                    help="Which layer neurons to describe, comma separated (no spaces), Pytorch module names")
parser.add_argument("--d_probe", type=str, default="broden")
parser.add_argument("--concept_set", type=str, default="data/20k.txt", help="Path to txt file containing concept set")
parser.add_argument("--batch_size", type=int, default=200, help="Batch size when running CLIP/target model")
parser.add_argument("--device", type=str, default="cuda", help="whether to use GPU/which gpu")
parser.add_argument("--activation_dir", type=str, default="saved_activations", help="where to save activations")
parser.add_argument("--result_dir", type=str, default="", help="where to save results")
parser.add_argument("--pool_mode", type=str, default="avg", help="Aggregation function for channels, max or avg")
parser.add_argument("--similarity_fn", type=str, default="soft_wpmi",
                    choices=["soft_wpmi", "wpmi", "rank_reorder", "cos_similarity", "cos_similarity_cubed"])


def main(argv=None):
    args = parser.parse_args(argv)
    args.target_layers = args.target_layers.split(",")
    setup_device(args)
    live = CLIP_og_utils.save_activations(clip_name=args.clip_model, target_name=args.target_model,
                                   target_layers=args.target_layers, d_probe=args.d_probe,
                                   concept_set=args.concept_set, batch_size=args.batch_size,
                                   device=args.device, pool_mode=args.pool_mode, save_dir=args.activation_dir)

    def names_for(layer):
        return CLIP_og_utils.get_save_names(clip_name=args.clip_model, target_name=args.target_model,
                                            target_layer=layer, d_probe=args.d_probe, concept_set=args.concept_set,
                                            pool_mode=args.pool_mode, save_dir=args.activation_dir)
    df = describe_layers(args, CLIP_og_utils, names_for, "clip", pass_top_k=False, pass_d_probe=False, live=live)
    return write_results(df, args, variant="clip", live=live)


if __name__ == '__main__':
    main()
