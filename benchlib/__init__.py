"""bench.py in parts: launch (rank start-up, torch-free), cpu (the CPU baseline), roofline (bytes, counters, ceilings),
extras (secondary measurements), headline (the timed whole-sphere R(Q) and the ONE line)."""
