"""The few-pairs scenario of tests/test_gpu_multi.py::test_multi_config3_shape_with_traceback_waves in a loop, with the end-of-batch
counters of the -DBA_ENDHIST build (offers made / slots refilled / slots taken over): python tools/dev/donate_dbg.py [rounds]"""
import os, sys, ctypes as C
os.environ.update(BA_FORCE_TB="1", BA_WGS_PER_CU="1", BA_FORCE_MULTI="1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from block_aligner_amd import hip as H, scores as S, synth
from oracle.oracle_py import Oracle
H.use_library(os.path.join(os.path.dirname(H.LIB_PATH), "libblock_aligner_hip_endhist.so"))
o = Oracle("avx2")
NUC = S.NucMatrix.new_simple(2, -3)
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
npairs = int(sys.argv[2]) if len(sys.argv) > 2 else 600
pairs = synth.make_pairs(npairs, (3000, 10000), (300, 1000), 500, synth.DNA, seed=4321)
ref = o.batch_align(NUC, pairs.pool, pairs.q_off, pairs.q_len, pairs.r_off, pairs.r_len, (-5, -1), (128, 1024), 100, ("trace", "x_drop"), cigar_eq=True, threads=16)
H.lib().ba_batch_prof.argtypes = [C.c_void_p, C.c_void_p]
for it in range(rounds):
    b = H.BatchAligner(NUC, (-5, -1), (128, 1024), 100, H.TRACE | H.X_DROP | H.CIGAR_EQ, pairs.pool, pairs.q_off, pairs.q_len, pairs.r_off, pairs.r_len)
    print("round", it, b.info()["kernel"], flush=True)
    b.run()
    res = b.results()
    prof = np.zeros(128, np.uint64); H.lib().ba_batch_prof(b._h, prof.ctypes.data)
    bad_s = np.nonzero(res["status"])[0]
    bad_sc = np.nonzero(res["score"] != ref["scores"])[0] if "scores" in ref else []
    bad_c = np.nonzero(res["cigar_len"] != ref["cig_len"])[0]
    print(f"  gave-up waves {int(prof[59])} (counted {int(prof[61])} of n_fill {int(prof[62])}; final by offering {int(prof[55])}, by idling {int(prof[54])}, idle entries {int(prof[53])}) offers {int(prof[56])} refills {int(prof[57])} steals {int(prof[58])}  status!=0 {bad_s[:5]} score {bad_sc[:5]} cigar_len {bad_c[:5]} retried {b.retried()}", flush=True)
    b.close()
