"""Does a big buffer's memory come back when its handle is destroyed?  (free device memory before / after cycles of a 2 GB basis)"""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lanczos_amd import _capi, synthetic
A = synthetic.laplacian_2d_5pt(1000, 1000)
M = A.shape[0]
v0 = np.ones(M) / np.sqrt(M)
h0 = _capi.Handle(0)
free0 = h0.device_memory()[0]
out = []
for cyc in range(6):
    h = _capi.Handle(0)
    h.set_csr(M, 0, A.rowptr, A.colidx, A.vals)
    h.run(250 + cyc, v0)   # 2 GB basis: a VMM range
    during = h0.device_memory()[0]
    h.close()
    out.append({"cycle": cyc, "held_during_GB": round((free0 - during) / 1e9, 2), "not_returned_GB": round((free0 - h0.device_memory()[0]) / 1e9, 2)})
print(json.dumps({"LZ_VMM_ADDRESS_REUSE": os.environ.get("LZ_VMM_ADDRESS_REUSE"), "LZ_NO_VMM": os.environ.get("LZ_NO_VMM"), "cycles": out}))
