"""calibration of SQ_VALU_MFMA_BUSY_CYCLES: the library's saturating v_mfma_f64_16x16x4 probe under the counter (run under rocprofv3 --pmc)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from qmps_amd import EnergyEngine
eng = EnergyEngine(16, 64)
for w in (1, 2, 4):
    print('waves per SIMD', w, 'TFLOP/s', eng.probe_fp64_mfma_tflops(w))
