"""Public model names of the HIP hot path; they match what the reference's tools import from its ``models`` package
(BIG-C for VidVRD and VidOR, the pairwise baseline, the grounding scorer)."""
from . import grd_model_v5 as _grounding
from . import model_0v7 as _vidor
from . import model_0v10 as _vidvrd
from . import model_pairwise_baseline as _pairwise

BIG_C_vidvrd = _vidvrd.BIG_C
BIG_C_vidor = _vidor.BIG_C
Base_C = _pairwise.Base_C
DEBUG = _grounding.DEBUG

__all__ = ["BIG_C_vidvrd", "BIG_C_vidor", "Base_C", "DEBUG"]
