"""Drop-in for the reference's concept_vit/describe_broad_neurons.py: Mammo-CLIP dissector (M-/C-Mammo-CLIP
Dissect), top-10 concepts per neuron (:101), `--top_k` (:39), output file-name ladder (:128-169).  The reader
uses the same prefix the writer used (the reference hard-codes `latest_..._mammo_pretrained_`, :90-92)."""
import argparse

from . import utils
from ._driver import broad_file_names, describe_layers, setup_device, write_results

parser = argparse.ArgumentParser(description='CLIP-Dissect')
parser.add_argument("--clip_model", type=str, default="ViT-B/16")
parser.add_argument("--num_class", type=int, default=1)
parser.add_argument("--target_model", type=str, default="breastclip")
parser.add_argument("--target_layers", type=str, default="image_encoder._blocks[0]")
parser.add_argument("--d_probe", type=str, default="vindr")
parser.add_argument("--concept_set", type=str, default="data/20k.txt")
parser.add_argument("--batch_size", type=int, default=200)
parser.add_argument("--device", type=str, default="cuda")
parser.add_argument("--activation_dir", type=str, default="saved_activations")
parser.add_argument("--result_dir", type=str, default="results")
parser.add_argument("--pool_mode", type=str, default="avg")
parser.add_argument("--top_k", type=int, default=100, help="Top-k activating images per neuron")
parser.add_argument("--similarity_fn", type=str, default="soft_wpmi",
                    choices=["soft_wpmi", "wpmi", "rank_reorder", "cos_similarity", "cos_similarity_cubed"])
parser.add_argument("--Breast_clip_chkpt", type=str, default=None, help="LOCAL path to a Mammo-CLIP checkpoint")
parser.add_argument("--finetuned_img_classifier_chkpt", type=str, default=None)
parser.add_argument("--arch", type=str, default="upmc_breast_clip_det_b5_period_n_ft")


def main(argv=None, prebuilt=None):
    """prebuilt: optional dict(clip_model=, target_model=, data=) -- models and the resident probe set of an earlier
    call (bench.py times repeated dissections without rebuilding them)."""
    args = parser.parse_args(argv)
    args.target_layers = [l.strip() for l in args.target_layers.split(",")]
    setup_device(args)
    live = utils.save_activations(clip_name=args.clip_model, target_name=args.target_model,
                                  target_layers=args.target_layers, d_probe=args.d_probe, concept_set=args.concept_set,
                                  batch_size=args.batch_size, device=args.device, pool_mode=args.pool_mode,
                                  save_dir=args.activation_dir, breast_clip_ckh=args.Breast_clip_chkpt,
                                  fine_tuned_ckh=args.finetuned_img_classifier_chkpt, args=args, prebuilt=prebuilt)
    pre = args.activation_dir + utils.save_prefix(args.d_probe, args.Breast_clip_chkpt,
                                                  args.finetuned_img_classifier_chkpt)

    def names_for(layer):
        t, c, x = utils.get_save_names(clip_name=args.clip_model, target_name=args.target_model, target_layer=layer,
                                       d_probe=args.d_probe, concept_set=args.concept_set, pool_mode=args.pool_mode,
                                       save_dir=args.activation_dir)
        return pre + t, pre + c, pre + x
    df = describe_layers(args, utils, names_for, "og", pass_top_k=True, pass_d_probe=True, live=live)
    csv_name, txt_name = broad_file_names(args)
    return write_results(df, args, csv_name, txt_name, variant="og", live=live)


if __name__ == '__main__':
    main()
