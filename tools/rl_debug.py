import sys; sys.path.insert(0, ".")
from strkit_amd.realign import realign_pairs
print(realign_pairs(["ACGTACGTAC"], ["TTTTACGTACGTACTTT"]), flush=True)
