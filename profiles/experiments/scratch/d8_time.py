import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools.other_configs import run
for D, B, solver in ((8, 96, 'direct'), (8, 768, 'direct'), (8, 65536, 'direct'), (16, 768, 'squaring')):
    print(json.dumps(run(D, B, solver, reps=200 if B <= 4096 else 30)), flush=True)
