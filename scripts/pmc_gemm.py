import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scripts.ubench_gemm as u
from tinyrecurrentunet_amd._lib import PRO_BNRELU
u.run(32064, 128, 128, 128, PRO_BNRELU, reps=2)
u.run(32064, 128, 192, 64, PRO_BNRELU, reps=2)
