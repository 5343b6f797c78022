"""Dev tool (no GPU): static instruction audit of the stand-alone map kernel (k_map) per phase, from the ISA with line tables:
  hipcc ... -gline-tables-only -S --cuda-device-only k_step.hip -o /tmp/k_step_g.s ;  python scripts/isa_audit_map.py /tmp/k_step_g.s
Every instruction of k_map is attributed to the source line its .loc names (inlined helpers included) and the lines to phases of
csrc/k_map.hip.  STATIC counts (one copy of each loop body): how many instructions a phase's code holds and of which kind - the
dynamic count per launch is the PMC figure (SQ_INSTS_VALU) quoted beside it in profiles/."""
import re, sys, collections
src = sys.argv[1] if len(sys.argv) > 1 else "/tmp/k_step_g.s"
txt = open(src).read()
files = {int(m.group(1)): m.group(2) for m in re.finditer(r'\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', txt)}
i = txt.find("_ZN4kmap5k_mapE")
i = txt.find("\n", txt.find(":", i))
j = txt.find(".end_amdhsa_kernel", i)
body = txt[i:j].split("\n")
# phases of k_map.hip by line range (map_body), helpers by function
import os
mapsrc = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "drl_graph_exploration_amd", "csrc", "k_map.hip")).read().split("\n")
def line_of(pat, start=0):
    for k in range(start, len(mapsrc)):
        if pat in mapsrc[k]:
            return k + 1
    raise KeyError(pat)
L = {k: line_of("DRLGX_PROF(S, %d)" % k) for k in (16, 40, 41, 17, 42, 43, 21, 19, 20)}
f_predict = (line_of("void predict_info("), line_of("// VirtualMap::covarianceIntersection2D"))
f_fuse = (line_of("void ci_fuse("), line_of("// Sum over the 64 lanes of a wave"))
ladder0 = line_of("// occupancy ladder (OccupancyMap.cpp:64-138)")
ladder1 = line_of("// VirtualMap::updateProbability: prob = sum over num_samples")
def phase(fn, ln):
    if fn != "k_map.hip":
        return None  # an inlined helper of another file (Pose2 algebra, libm, reciprocals): goes to the phase of the code around it
    if f_predict[0] <= ln < f_predict[1]: return "A  EKF push-through (predict_info / predict_cell)"
    if f_fuse[0] <= ln < f_fuse[1]: return "C  covariance-intersection fusion (ci_fuse)"
    if ln < L[16]: return "prologue (carve, counts)"
    if ln < L[40]: return "tables: loads, clears"
    if ln < L[41]: return "landmark cells, pose windows, LLT of the pose information"
    if ln < L[17]: return "bbox sweep (narrow sensors only)"
    if ln < L[43]: return "A  range / FOV tests + compaction"
    if ln < L[21]: return "A  pair loop around the push-through (stage stores, masks)"
    if ladder0 <= ln < ladder1: return "C  occupancy ladder"
    if ln < L[19]: return "C  cell pass: tile walk, chain walk, outputs, utility terms"
    return "R  block reduction, outputs"
cnt = collections.OrderedDict()
cur = ("?", 0)
last_phase = "prologue (carve, counts)"
for l in body:
    m = re.match(r"\s+\.loc\s+(\d+)\s+(\d+)", l)
    if m:
        cur = (files.get(int(m.group(1)), "?"), int(m.group(2)))
        continue
    t = l.strip()
    if not l.startswith("\t") or not t or t[0] in ".;" or t.endswith(":"):
        continue
    op = t.split()[0]
    kind = ("fp64 VALU" if "_f64" in op else "other VALU" if op.startswith("v_") else "LDS" if op.startswith("ds_") else
            "SALU / branch" if op.startswith("s_") else "global / flat" if op.startswith(("global_", "flat_", "buffer_", "scratch_")) else "other")
    p = phase(*cur)
    if p is None:
        p = last_phase
    last_phase = p
    d = cnt.setdefault(p, collections.Counter())
    d[kind] += 1
    d["all"] += 1
kinds = ["all", "fp64 VALU", "other VALU", "SALU / branch", "LDS", "global / flat"]
print("%-92s" % "phase (static instructions of k_map, gfx950)" + "".join("%15s" % k for k in kinds))
tot = collections.Counter()
for p, d in sorted(cnt.items()):
    print("%-92s" % p + "".join("%15d" % d[k] for k in kinds))
    tot.update(d)
print("%-92s" % "total" + "".join("%15d" % tot[k] for k in kinds))
