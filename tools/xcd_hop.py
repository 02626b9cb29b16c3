"""What a hand-off between two workgroups costs on this box: inside one XCD through its L2 without
cache maintenance (mode 0), across XCDs behind agent-scope release / acquire (1), the same fences
inside one XCD (2) -- and an empty launch for scale.  docs/LABBOOK.md round 6, C2 item."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_quadrature_amd import Engine  # noqa: E402

e = Engine(0)
print("launch boundary (empty kernels back to back): %.2f us" % e.probe_launch(2000))
for kib in (1, 8, 32):
    for mode, name in ((0, "one XCD, sc1 flag + sc1 loads"), (1, "two XCDs, release / acquire"),
                       (2, "one XCD, release / acquire")):
        for rep in range(2):
            ns, xcc, bad = e.probe_xcd_hop(mode, 3000, kib)
            pairs = [(int(xcc[2 * p]), int(xcc[2 * p + 1])) if mode == 1 else
                     (int(xcc[p]), int(xcc[p + 8])) for p in range(8)]
            same = sum(a == b for a, b in pairs)
        print("payload %2d KiB, %-34s %7.0f ns per hop   stale words %d   pairs on one XCD %d / 8"
              % (kib, name + ":", ns, bad, same), flush=True)
e.close()
