"""Per-kernel register / LDS table from `make asm`'s resource_usage_*.txt (build/asm)."""
import re, sys, glob, os
d = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(__file__), "..", "build", "asm")
for f in sorted(glob.glob(os.path.join(d, "resource_usage_*.txt"))):
    cur = {}
    for line in open(f):
        m = re.search(r"remark: [^ ]+ (.*?): (.*?) \[-Rpass", line) or re.search(r"remark: .*?:\d+:\d+: (.*?): (.*?) \[-Rpass", line)
        if not m:
            continue
        k, v = m.group(1).strip(), m.group(2).strip()
        if k == "Function Name":
            cur = {"name": v}
        cur[k] = v
        if k.startswith("LDS Size"):
            print("%-28s %-60s vgpr=%-4s agpr=%-4s sgpr=%-4s scratch=%-5s occ=%-2s lds=%s" % (
                os.path.basename(f)[15:-4], cur["name"][:60], cur.get("VGPRs"), cur.get("AGPRs"), cur.get("SGPRs"),
                cur.get("ScratchSize [bytes/lane]"), cur.get("Occupancy [waves/SIMD]"), v))
