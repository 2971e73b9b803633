import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, oracle
from innr_amd import KNN_EXACT, KNN_MFMA
from innr_amd import batch as B
rows = oracle.generate_uniform(4000, 16, 3)
rows[1234, 5] = np.inf
q = oracle.generate_uniform(1, 16, 8)[0]; q[5] = 0.0
vb = B.VerticalBatch.from_rows(rows)
s = B.batch_dot(q, vb); print("batch_dot bits at 1234:", hex(int(s[1234:1235].view(np.uint32)[0])))
c = B.batch_cosine(oracle.generate_uniform(1, 16, 9)[0], vb, B.batch_norms(vb)); print("batch_cosine bits at 1234:", hex(int(c[1234:1235].view(np.uint32)[0])))
for e in (KNN_EXACT, KNN_MFMA):
    i, sc = B.batch_knn_dot_multi(q.reshape(1, -1), vb, 4000, engine=e)
    pos = int(np.where(i[0] == 1234)[0][0]); print("engine", e, "dot rank of 1234:", pos, hex(int(sc[0][pos:pos+1].view(np.uint32)[0])))
    i, sc = B.batch_knn_cosine_multi(oracle.generate_uniform(1, 16, 9), vb, 4000, engine=e)
    pos = int(np.where(i[0] == 1234)[0][0]); print("engine", e, "cos rank of 1234:", pos, hex(int(sc[0][pos:pos+1].view(np.uint32)[0])))
    i, sc = B.batch_knn_dot_multi(q.reshape(1, -1), vb, 6, engine=e); print("k=6 dot", i[0], [hex(int(x)) for x in sc[0].view(np.uint32)])
o = oracle.batch_dot(q, oracle.from_rows(rows)); print("oracle (this host) bits:", hex(int(o[1234:1235].view(np.uint32)[0])))
