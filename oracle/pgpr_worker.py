"""TEST INFRASTRUCTURE (CPU oracle), run as a child process by bench_pgpr.py's cpu_baseline leg only: one worker of the
multi-core CPU baseline of the PGPR beam search.  Loads the graph / tables / policy weights the bench saved, runs
oracle.pgpr's beam search (pgpr.py:440-484 restated) for its users and prints {"paths": ..., "seconds": ...}.

    python -m oracle.pgpr_worker <state.npz> <hops, e.g. 64,64,64> <user id> [<user id> ...]
"""

from __future__ import annotations

import json
import sys
import time

import numpy as np

from oracle import pgpr as op


def main(argv):
    z = np.load(argv[1])
    hops = [int(x) for x in argv[2].split(",")]
    users = np.asarray([int(x) for x in argv[3:]], np.int64)
    g = op.CKG(int(z["n_users"]), int(z["n_entities"]), int(z["ui_relation"]), z["inter_users"], z["inter_items"], z["heads"],
               z["relations"], z["tails"])
    W = {k[2:]: z[k] for k in z.files if k.startswith("w.")}
    orc = op.PGPROracle(g, int(z["n_items"]), z["user_emb"], z["entity_emb"], z["relation_emb"],
                        np.zeros((int(z["n_users"]), 1), np.int64), W, int(z["max_acts"]), int(z["state_history"]), False, (10,))
    t0 = time.perf_counter()
    _, _, _, trace = op.safe_leaves(orc, users, hops, with_trace=True)
    dt = time.perf_counter() - t0
    n = sum(sum(min(k, len(a)) for a in t["acts"]) for t, k in zip(trace, hops))
    print(json.dumps({"paths": int(n), "seconds": dt, "users": len(users)}))


if __name__ == "__main__":
    main(sys.argv)
