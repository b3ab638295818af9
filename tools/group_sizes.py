#!/usr/bin/env python3
"""How the suffixes that are still tied after the initial sort of C3 (12 symbols) spread over group sizes -- on the CPU, from the
text alone (64-bit hashes of all 12-byte windows, sorted).  Informs k_group_sort's size classes and the share that goes through
the global sort (groups of more than 1 024 members).  python tools/group_sizes.py   (about a minute, 6 GB)"""
import numpy as np, time, sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from suffix_array_amd import corpus
t0=time.time()
T=corpus.workload("c3_english_256m")
n=T.size
print("text", n, time.time()-t0, flush=True)
D=12
# 64-bit hash of every 12-byte window (windows running past the end are ignored: negligible)
h=np.zeros(n-D+1,dtype=np.uint64)
M=np.uint64(0x100000001b3)
for j in range(D):
    h*=M
    h+=T[j:n-D+1+j].astype(np.uint64)+np.uint64(1)
print("hashed", time.time()-t0, flush=True)
h.sort()
print("sorted", time.time()-t0, flush=True)
b=np.flatnonzero(np.concatenate(([True],h[1:]!=h[:-1],[True])))
sz=np.diff(b)
print("groups", sz.size, "tied members", int(sz[sz>1].sum()), flush=True)
edges=[2,3,5,9,17,33,65,129,257,513,1025,2049,4097,8193,16385,65537,262145,1<<30]
lo=2
tot=int(sz[sz>1].sum())
for e in edges[1:]:
    m=(sz>=lo)&(sz<e)
    print("size [%d, %d): groups %d members %d (%.1f%% of tied)"%(lo,e,int(m.sum()),int(sz[m].sum()),100.0*sz[m].sum()/tot))
    lo=e
