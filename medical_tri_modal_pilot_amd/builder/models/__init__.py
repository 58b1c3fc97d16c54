"""Model registry with the reference's contract (builder/models/__init__.py:14-51):
``get_model(args)`` imports ``builder.models.8_missing_models.<args.model>`` and returns the
class named ``<args.model>.upper()``; the caller constructs it as ``Model(args)``.

Built models: tri_mbt_vsltcls (the benchmarked one), tri_mbt_vsltcls_noshareumse, tri_mbt_v1, tri_mbt_v2,
tri_mbt_vnoshavgtr, tri_mbt_vflexible / 2 / 3, bi_vslttxt_mbt_v1, bitxt_mbt_vflexible1, biimg_mbt_vflexible1, bi_vsltimg_mbt_v1 -- reference-equivalent
for inference AND training (bi_vsltimg_mbt_v1, tri_mbt_v2 and tri_mbt_vnoshavgtr train the Swin-T image encoder, as their references do)."""
import importlib


def get_model(args):
    try:
        module = importlib.import_module(__name__ + ".8_missing_models." + args.model)
    except ModuleNotFoundError as e:
        raise NotImplementedError(
            f"model '{args.model}' is not part of the MI355X hot path (built: tri_mbt_vsltcls, tri_mbt_vsltcls_noshareumse, tri_mbt_v1, tri_mbt_v2, tri_mbt_vnoshavgtr, tri_mbt_vflexible, tri_mbt_vflexible2, tri_mbt_vflexible3, bi_vslttxt_mbt_v1, bitxt_mbt_vflexible1, bi_vsltimg_mbt_v1, biimg_mbt_vflexible1)") from e
    return getattr(module, args.model.upper())
