"""One rank of tests/test_gpu_boundary_r2.py::test_rccl_gather_two_ranks: camera `rank` on GPU `rank`, EgoLanes label maps
all-gathered through the C ABI (vp_comm_* / vp_gather); the RCCL unique id travels through a file, the way a C++ host
without torch would hand it over.  usage: gather_rank.py RANK WORLD ID_FILE OUT_NPY"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def main():
    rank, world, idf, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    from autoware_vision_pilot_amd import lib, synthetic, weights as vw

    if rank == 0:
        uid = lib.Comm.unique_id()
        with open(idf + ".tmp", "wb") as f:
            f.write(uid)
        os.replace(idf + ".tmp", idf)
    else:
        t0 = time.time()
        while not os.path.exists(idf):
            if time.time() - t0 > 300:
                raise SystemExit("no unique id from rank 0")
            time.sleep(0.05)
        uid = open(idf, "rb").read()
    eng = lib.Engine("egolanes", vw.pack_state_dict(synthetic.make_state_dict("egolanes", 2)), precision="fp16", gpu_id=rank)
    eng.set_decode_mode(lib.VP_DECODE_LANE_LABEL)
    eng.upload_frame(synthetic.synthetic_frame(720, 1280, 10 + rank))
    comm = lib.Comm(uid, rank, world, rank, 80 * 160)
    for _ in range(3):
        eng.enqueue()
        comm.gather(eng, lib.VP_GATHER_MASK)
    got = comm.fetch(eng)
    eng.fetch_outputs()
    assert np.array_equal(got[rank].reshape(80, 160), eng.mask())
    np.save(out, got)
    comm.close()
    eng.close()


if __name__ == "__main__":
    main()
