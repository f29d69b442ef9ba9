"""`cm3p.configuration_cm3p` served by the MI355X build (see cm3p_amd/configuration_cm3p.py)."""
from cm3p_amd.configuration_cm3p import *  # noqa: F401,F403
from cm3p_amd.configuration_cm3p import __all__  # noqa: F401
