"""build_model(args, load_config_dict) -> (nn.Module, args) -- the drop-in factory.

Same signature, dict keys, defaults and return value as the reference
(models/build_model.py:9-48): reads vocab_size, n_layer, n_head, d_model,
d_inner, dropout, d_condition, conditioning (+regression, overwrite_dropout),
forces max_seq=2048 and pad_token=0, drops d_condition for continuous_token; regression=True builds the
evaluation model MusicRegression (output_size 2; forward only here).
One extra optional key, `compute_dtype` ("bf16" default, "fp32" = exact-f32
parity tier), selects the storage type of the HIP engine.
"""
from .music_transformer import MusicRegression, MusicTransformerContinuousToken, MusicTransformerMulti


def set_dropout(model, rate):
    """models/build_model.py:2-7.  The engine keeps a single dropout rate (all four dropout
    sites of the reference share args.dropout)."""
    model.dropout_p = float(rate)
    return model


def build_model(args, load_config_dict=None):
    if load_config_dict is not None:
        args = load_config_dict

    config = {
        "vocab_size": args["vocab_size"],
        "num_layer": args["n_layer"],
        "num_head": args["n_head"],
        "embedding_dim": args["d_model"],
        "d_inner": args["d_inner"],
        "dropout": args["dropout"],
        "d_condition": args["d_condition"],
        "max_seq": 2048,
        "pad_token": 0,
        "compute_dtype": args.get("compute_dtype", "bf16"),
    }

    if "regression" not in args:
        args["regression"] = False

    if args["regression"]:
        config["output_size"] = 2                       # build_model.py:29-32; inference only in this build
        model = MusicRegression(**config)
    elif args["conditioning"] == "continuous_token":
        del config["d_condition"]
        model = MusicTransformerContinuousToken(**config)
    else:
        model = MusicTransformerMulti(**config)

    if load_config_dict is not None and args is not None:
        if args.get("overwrite_dropout", False):
            model = set_dropout(model, args["dropout"])
            print(f"Dropout rate changed to {args['dropout']}")
    return model, args
