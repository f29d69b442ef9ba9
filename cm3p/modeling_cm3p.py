"""`cm3p.modeling_cm3p` served by the MI355X build (see cm3p_amd/modeling_cm3p.py)."""
from cm3p_amd.modeling_cm3p import *  # noqa: F401,F403
from cm3p_amd.modeling_cm3p import (  # noqa: F401
    CM3PAudioModelOutput, CM3PBeatmapModelOutput, CM3PMetadataModelOutput, __all__)
