"""Importable pieces of bench.py (the workload tables, the CPU baseline leg, the superpoint-stage workload, the multi-rank
launcher)."""
