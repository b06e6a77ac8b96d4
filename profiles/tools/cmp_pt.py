"""Compare two dumps of outputs (lists of lists of tensors) bit for bit: python profiles/tools/cmp_pt.py A.pt B.pt"""
import sys
import torch
a, b = torch.load(sys.argv[1]), torch.load(sys.argv[2])
bad = [i for i, (x, y) in enumerate(zip(a, b)) if any(not torch.equal(p, q) for p, q in zip(x, y))]
print("cases", len(a), "different:", bad)
