from .encoders.modules import GeneralConditioner  # noqa: F401  (reference: sgm/modules/__init__.py)

UNCONDITIONAL_CONFIG = {
    "target": "v3d_amd.sgm.modules.GeneralConditioner",
    "params": {"emb_models": []},
}
