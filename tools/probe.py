"""Measure the box's FP64 VALU / FP64 MFMA / HBM rates with libqmps_hip's probes (one JSON line)."""
import json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qmps_amd import EnergyEngine, _lib
eng = EnergyEngine(4, 1024)
out = {'device': _lib.device_info(0), 'fp64_valu_tflops': eng.probe_fp64_tflops(),
       'fp64_mfma_tflops': {w: eng.probe_fp64_mfma_tflops(w) for w in (1, 2, 4)}, 'hbm_copy_gbps': eng.probe_hbm_gbps()}
print(json.dumps(out))
