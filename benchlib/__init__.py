"""The workloads behind bench.py (repo root): common helpers, `--workload overlap | evolve | rotosolve`; the headline energy workload is bench.py itself."""
