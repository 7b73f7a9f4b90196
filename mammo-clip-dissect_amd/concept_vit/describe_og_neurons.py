"""Drop-in for the reference's concept_vit/describe_og_neurons.py: CLIP dissector, Mammo-CLIP (or other)
target, top-10 concepts per neuron (:99).  Flags as the reference (:14-47); wandb logging is optional (absent
here) and CUDA_LAUNCH_BLOCKING is not forced."""
import argparse

from . import og_utils
from ._driver import describe_layers, setup_device, write_results

parser = argparse.ArgumentParser(description='CLIP-Dissect')
parser.add_argument("--clip_model", type=str, default="ViT-B/16",
                    choices=['RN50', 'RN101', 'RN50x4', 'RN50x16', 'RN50x64', 'ViT-B/32', 'ViT-B/16', 'ViT-L/14'])
parser.add_argument("--num_class", type=int, default=1, help="Number of classes in the classifier")
parser.add_argument("--target_model", type=str, default="breastclip", help="Which model to dissect")
parser.add_argument("--target_layers", type=str, default="image_encoder._blocks[0]")
parser.add_argument("--d_probe", type=str, default="imagenet_subsets")
parser.add_argument("--concept_set", type=str, default="data/20k.txt")
parser.add_argument("--batch_size", type=int, default=200)
parser.add_argument("--device", type=str, default="cuda")
parser.add_argument("--activation_dir", type=str, default="saved_activations")
parser.add_argument("--result_dir", type=str, default="results")
parser.add_argument("--pool_mode", type=str, default="avg")
parser.add_argument("--similarity_fn", type=str, default="soft_wpmi",
                    choices=["soft_wpmi", "wpmi", "rank_reorder", "cos_similarity", "cos_similarity_cubed"])
parser.add_argument("--Breast_clip_chkpt", type=str, default=None, help="LOCAL path to a Mammo-CLIP checkpoint")
parser.add_argument("--finetuned_img_classifier_chkpt", type=str, default=None)
parser.add_argument("--arch", type=str, default="upmc_breast_clip_det_b5_period_n_ft")


def main(argv=None):
    args = parser.parse_args(argv)
    args.target_layers = [l.strip() for l in args.target_layers.split(",")]
    setup_device(args)
    live = og_utils.save_activations(clip_name=args.clip_model, target_name=args.target_model,
                              target_layers=args.target_layers, d_probe=args.d_probe, concept_set=args.concept_set,
                              batch_size=args.batch_size, device=args.device, pool_mode=args.pool_mode,
                              save_dir=args.activation_dir, breast_clip_ckh=args.Breast_clip_chkpt,
                              fine_tuned_ckh=args.finetuned_img_classifier_chkpt, args=args)
    pre = args.activation_dir + og_utils.save_prefix(args.target_model, args.d_probe, args.Breast_clip_chkpt,
                                                     args.finetuned_img_classifier_chkpt)

    def names_for(layer):
        t, c, x = og_utils.get_save_names(clip_name=args.clip_model, target_name=args.target_model,
                                          target_layer=layer, d_probe=args.d_probe, concept_set=args.concept_set,
                                          pool_mode=args.pool_mode, save_dir=args.activation_dir)
        return pre + t, pre + c, pre + x
    df = describe_layers(args, og_utils, names_for, "og", pass_top_k=False, pass_d_probe=True, live=live)
    return write_results(df, args, "descriptions.csv", "args.txt", variant="og", live=live)


if __name__ == '__main__':
    main()
