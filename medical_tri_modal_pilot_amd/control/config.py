"""Flag surface of the reference's ``control/config.py:10-159`` (every flag keeps its name, type,
default and choices -- README.md:44 runs unchanged), plus additive build-only flags.

Unlike the reference this module does not parse ``sys.argv`` at import: ``build_parser()`` /
``parse_args(argv)`` are explicit, and ``from ...control.config import args`` still works through
a lazy module attribute (parsed from sys.argv on first access), which is what ``2_train.py:22`` does.
"""
import argparse
import os

_VITALS = ['HR', 'RR', 'BT', 'SBP', 'DBP', 'Sat', 'Hematocrit', 'PLT', 'WBC', 'Bilirubin', 'pH', 'HCO3',
           'Creatinine', 'Lactate', 'Potassium', 'Sodium']

# (flags, kwargs) -- order and values follow control/config.py of the reference
_FLAGS = [
    (("--seed",), dict(type=int, default=0)),
    (("--seed-list",), dict(type=list, default=[412, 1004, 2023])),
    (("--device",), dict(type=int, default=1, nargs="+")),
    (("--cpu",), dict(type=int, default=0)),
    (("--num-workers",), dict(type=int, default=5)),
    (("--gpus",), dict(type=int, default=1)),
    (("--reset",), dict(default=False, action="store_true")),
    (("--project-name",), dict(type=str, default="small1")),
    (("--checkpoint", "-cp"), dict(type=bool, default=False)),
    (("--flexconst",), dict(type=float, default=1)),
    (("--prediction-range",), dict(type=int, default=12)),
    (("--min-inputlen",), dict(type=int, default=3)),
    (("--window-size",), dict(type=int, default=24)),
    (("--vslt-type",), dict(type=str, default="TIE", choices=["carryforward", "TIE", "QIE"])),
    (("--realtime",), dict(type=int, default=1, choices=[0, 1])),
    (("--multiimages",), dict(type=int, default=0, choices=[0, 1])),
    (("--TIE-len",), dict(type=int, default=1000)),
    (("--ar-lowerbound",), dict(type=float, default=0.7)),
    (("--ar-upperbound",), dict(type=float, default=1.3)),
    (("--input-types",), dict(type=str, default="vslt", choices=["vslt", "vslt_img", "vslt_txt", "vslt_img_txt"])),
    (("--output-type",), dict(type=str, default="mortality", choices=['mortality', 'vasso', 'intubation', 'cpr', 'transfer'])),
    (("--predict-type",), dict(type=str, default="within", choices=["within", "multi_task_within", "multi_task_range", "seq_pretrain"])),
    (("--modality-inclusion",), dict(type=str, default="train-full_test-full",
                                     choices=['train-full_test-full', 'train-missing_test-missing', 'train-full_test-missing'])),
    (("--fullmodal-definition",), dict(type=str, default="txt1_img1", choices=["txt1_img1", "img1", "txt1"])),
    (("--train-data-path",), dict(type=str, default="./data/sample_data/train")),
    (("--test-data-path",), dict(type=str, default="./data/sample_data/test")),
    (("--dir-result",), dict(type=str, default="/mnt/aitrics_ext/ext01/destin/multimodal/mlhc_final_models")),
    (("--image-data-path",), dict(type=str, default="/home/claire/")),
    (("--cross-fold-val",), dict(type=int, default=0, choices=[1, 0])),
    (("--val-data-ratio",), dict(type=float, default=0.1)),
    (("--imgtxt-time",), dict(type=int, default=0, choices=[0, 1])),
    (("--missing-exhaustive",), dict(type=int, default=0, choices=[0, 1])),
    (("--epochs",), dict(type=int, default=50)),
    (("--batch-size",), dict(type=int, default=32)),
    (("--l2-coeff",), dict(type=float, default=0.002)),
    (("--dropout",), dict(type=float, default=0.1)),
    (("--activation",), dict(choices=['selu', 'relu'], default='relu', type=str)),
    (("--optim",), dict(type=str, default='adamw', choices=['sgd', 'sgd_lars', 'adam', 'adam_lars', 'adamw', 'adamw_lars'])),
    (("--lr-scheduler",), dict(type=str, default="CosineAnnealing", choices=["CosineAnnealing", "Single"])),
    (("--lr-init",), dict(type=float, default=1e-3)),
    (("--t_0", "-tz"), dict(type=int, default=50)),
    (("--t_mult", "-tm"), dict(type=int, default=2)),
    (("--t_up", "-tup"), dict(type=int, default=5)),
    (("--gamma", "-gam"), dict(type=float, default=0.5)),
    (("--momentum", "-mo"), dict(type=float, default=0.9)),
    (("--weight_decay", "-wd"), dict(type=float, default=1e-6)),
    (("--patient-time",), dict(default=False)),
    (("--threshold",), dict(type=float, default=0.5)),
    (("--output-dim",), dict(type=int, default=1)),
    (("--txt-num-layers",), dict(type=int, default=8)),
    (("--txt-dropout",), dict(type=float, default=0.1)),
    (("--txt-model-dim",), dict(type=int, default=256)),
    (("--txt-num-heads",), dict(type=int, default=4)),
    (("--txt-classifier-nodes",), dict(type=int, default=64)),
    (("--txt-tokenization",), dict(type=str, default="bert", choices=["word", "character", "bpe", "bert"])),
    (("--berttype",), dict(type=str, default="biobert", choices=["biobert", "bert"])),
    (("--biobert-path",), dict(type=str, default="./data/mimic4_embeddings.h5",
                               choices=["./data/mimic4_embeddings.h5", "./data/mimic4_clstoken.h5"])),
    (("--character-token-max-length",), dict(type=int, default=512)),
    (("--word-token-max-length",), dict(type=int, default=128)),
    (("--bpe-token-max-length",), dict(type=int, default=256)),
    (("--bert-token-max-length",), dict(type=int, default=128)),
    (("--enc-depth",), dict(type=int, default=3, choices=[1, 2, 3])),
    (("--hidden-size",), dict(type=int, default=256)),
    (("--transformer-dim",), dict(type=int, default=256)),
    (("--transformer-num-layers",), dict(type=int, default=6)),
    (("--transformer-num-head",), dict(type=int, default=4)),
    (("--resnet-num-layers",), dict(type=int, default=18, choices=[18, 34, 50])),
    (("--vit-num-layers",), dict(type=int, default=8, choices=[4, 8, 10, 12])),
    (("--vit-patch-size",), dict(type=int, default=16, choices=[8, 16])),
    (("--img-model-type",), dict(type=str, default="swin", choices=["resnet18", "resnet50", "swin", "vit", "maxvit"])),
    (("--img-pretrain",), dict(type=str, default="Yes", choices=["No", "Yes"])),
    (("--image-size",), dict(type=int, default=224, choices=[224, 512])),
    (("--image-train-type",), dict(type=str, default="resize_affine_crop",
                                   choices=["random", "resize", "resize_crop", "resize_affine_crop", "randaug"])),
    (("--image-test-type",), dict(type=str, default="resize_crop", choices=["center", "resize", "resize_crop", "resize_larger"])),
    (("--image-norm-type",), dict(type=str, default="HE", choices=["HE", "CLAHE"])),
    (("--residual-bottlenecks",), dict(type=int, default=0, choices=[0, 1])),
    (("--mbt-bottlenecks-n",), dict(type=int, default=4)),
    (("--mbt-fusion-startIdx",), dict(type=int, default=0)),
    (("--mbt-only-vslt",), dict(type=int, default=0)),
    (("--model-types",), dict(type=str, default="detection", choices=["detection", "classification"])),
    (("--loss-types",), dict(type=str, default="bce", choices=["bceandsoftmax", "softmax", "bces", "bce", "wkappa", "rmse"])),
    (("--auxiliary-loss-input",), dict(type=str, default=None, choices=[None, "directInput", "encOutput"])),
    (("--auxiliary-loss-type",), dict(type=str, default="None", choices=["None", "rmse", "tdecoder", "tdecoder_rmse"])),
    (("--auxiliary-loss-weight",), dict(type=float, default=1.0)),
    (("--mandatory-vitalsign-labtest",), dict(type=list, default=['HR', 'RR', 'BT', 'SBP', 'DBP', 'Sat'])),
    (("--vitalsign-labtest",), dict(type=list, default=_VITALS)),
    (("--model",), dict(type=str, default="gru_d")),
    (("--log-iter",), dict(type=int, default=10)),
    (("--nonPatNegSampleN",), dict(type=int, default=4)),
    (("--PatNegSampleN",), dict(type=int, default=1)),
    (("--PatPosSampleN",), dict(type=int, default=5)),
    (("--best",), dict(default=True, action="store_true")),
    (("--last",), dict(default=False, action="store_true")),
    (("--fuse-baseline",), dict(type=str, default=None, choices=["Medfuse", "MMTM", "DAFT", "Retain", "Multi"])),
    (("--mmtm-ratio",), dict(type=float, default=4)),
    (("--daft_activation",), dict(type=str, default='linear')),
    (("--fusion-type",), dict(type=str, default='fused_ehr')),
    (("--image-observed-prop",), dict(type=int, default=100, choices=[10, 30, 50, 70, 90, 100])),
    (("--text-observed-prop",), dict(type=int, default=100, choices=[10, 30, 50, 70, 90, 100])),
]

# build-only additions (never rename or shadow a reference flag)
_BUILD_FLAGS = [
    (("--compute-dtype",), dict(type=str, default="bf16", choices=["bf16", "fp32"],
                                help="bf16: MFMA performance build; fp32: parity build (exact fp32 MFMA)")),
    (("--ddp",), dict(type=int, default=0, choices=[0, 1], help="data-parallel training over RCCL (one rank per GPU)")),
    (("--fused-adamw",), dict(type=int, default=1, choices=[0, 1], help="mtmp_adamw_step over flat buffers")),
    (("--hip-graph",), dict(type=int, default=1, choices=[0, 1],
                            help="replay zero_grad+forward+backward of the training step from a captured hipGraph "
                                 "(pays off when the host cannot enqueue ~770 launches per step fast enough)")),
    (("--hip-graph-fallback",), dict(type=int, default=0, choices=[0, 1],
                                     help="1: a failed hipGraph capture falls back to eager launches with a warning "
                                          "(default: it raises -- the eager step is host-bound)")),
    (("--hip-graph-max",), dict(type=int, default=12,
                                help="input shapes (length buckets) a trainer captures at most; further shapes run as eager "
                                     "launches (captured graphs cannot be released on ROCm 7.2: graph.py)")),
    (("--graph-stages",), dict(type=int, default=0,
                               help="number of hipGraphs the captured step is cut into at fusion-layer boundaries "
                                    "(k > 0: k even groups of layers; 0 = auto: one graph, under --ddp 1 two, cut behind the first fusion layer, so "
                                    "that the all-reduce of the later layers' gradients overlaps the rest of the backward)")),
    (("--pack-rows",), dict(type=int, default=1, choices=[0, 1],
                            help="1: the vital-sign stream runs through the fusion layers PACKED -- its samples' valid rows "
                                 "back to back, no pad rows -- whenever the bf16 kernels and the model allow it (results do "
                                 "not depend on pad rows); 0: the reference's padded [B, T] layout")),
    (("--skip-missing-images",), dict(type=int, default=1, choices=[0, 1],
                                      help="1: the frozen image encoder runs on the samples that HAVE an image only (the others' "
                                           "features are read by nothing); 0: a zero image goes through the encoder as in the "
                                           "reference")),
    (("--n-images",), dict(type=int, default=3, help="images per sample when --multiimages 1 (reference: 3)")),
    (("--synthetic",), dict(type=int, default=0, choices=[0, 1], help="train on synthetic batches (SURVEY.md §8d)")),
]


def build_parser() -> argparse.ArgumentParser:
    parser = argparse.ArgumentParser()
    for names, kw in _FLAGS + _BUILD_FLAGS:
        parser.add_argument(*names, **kw)
    return parser


def parse_args(argv=None):
    a = build_parser().parse_args(argv)
    a.dir_root = os.getcwd()
    if "train-full" in a.modality_inclusion:          # same guard as config.py:157-159
        need = [t + "1" for t in a.input_types.split("_") if t != "vslt"]
        if not all(n in a.fullmodal_definition.split("_") for n in need):
            raise ValueError('invalid input_types for full_modal with fullmodal_definition!!!')
    return a


_args = None


def __getattr__(name):
    global _args
    if name == "args":
        if _args is None:
            _args = parse_args()
        return _args
    raise AttributeError(name)
