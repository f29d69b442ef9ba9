"""cm3p_amd: MI355X-native implementation of CM3P's contrastive training hot path.

Hand-written HIP kernels (cm3p_amd/csrc, built into libcm3p_hip.so) behind a C ABI (include/cm3p_hip.h), with a
Python host side that mirrors the reference's `cm3p.modeling_cm3p` / `cm3p.configuration_cm3p` surface.
There is no CPU or PyTorch-op fallback: without the library and a GPU the model raises.
"""
__version__ = "0.1.0"


def __getattr__(name):
    # lazy: importing the package must not need transformers/torch until a model class is asked for
    if name in ("CM3PConfig", "CM3PMetadataConfig", "CM3PAudioConfig", "CM3PBeatmapConfig"):
        from . import configuration_cm3p as m

        return getattr(m, name)
    if name in ("CM3PModel", "CM3POutput", "CM3PPreTrainedModel", "CM3PBeatmapModel", "CM3PMetadataModel",
                "CM3PBeatmapTransformer", "CM3PMetadataTransformer", "CM3PAudioEncoder"):
        from . import modeling_cm3p as m

        return getattr(m, name)
    if name == "Muon":
        from .muon import Muon

        return Muon
    raise AttributeError(name)
