"""Drop-in names of the reference's ``models/__init__.py:1-4`` for the HIP hot path."""
from .model_0v10 import BIG_C as BIG_C_vidvrd
from .model_0v7 import BIG_C as BIG_C_vidor
from .grd_model_v5 import DEBUG
from .model_pairwise_baseline import Base_C
