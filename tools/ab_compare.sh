#!/bin/bash
# Interleaved A/B of two builds of libsf_hip.so ON ONE GPU BOX (the package runs at its power limit: clocks differ from box to box and
# fall over the first minute of a run, so only runs that alternate on the same box compare).
#   tools/build_variant.sh <tag> "<-D flags>"          builds staticfusion_amd/csrc/libsf_hip_<tag>.so here (it travels with gpurun)
#   gpurun --timeout 1200 -- 'bash tools/ab_compare.sh libsf_hip.so libsf_hip_<tag>.so [reps=3] [batch=5120] [stage ...]'
# prints frames/s of both workloads of the bench line per repetition and the mean ratio B / A; extra arguments are names of
# in-kernel stage timers (tools/stage_profile.py) to print beside them, e.g. warp residuals.
set -u
cd "$(dirname "$0")/.."
A=$1; B=$2; REPS=${3:-3}; BATCH=${4:-5120}; shift; shift; shift 2>/dev/null; shift 2>/dev/null
STAGES="$*"
C=$PWD/staticfusion_amd/csrc
TMP=$(mktemp)
for r in $(seq 1 $REPS); do
  for w in sphere static; do
    if [ $((r % 2)) -eq 1 ]; then order="$A $B"; else order="$B $A"; fi  # alternate who goes first: the second run of a pair is the warmer one
    for lib in $order; do
      out=$(SF_HIP_LIB=$C/$lib timeout -k 10 300 python tools/stage_profile.py --batch $BATCH --workload $w --steps 10)
      fps=$(echo "$out" | grep workload | sed -E 's/.* ([0-9]+) frames\/s.*/\1/')
      extra=""
      for s in $STAGES; do extra="$extra $s=$(echo "$out" | grep -E "^ +$s " | awk '{print $2}')us"; done
      echo "rep $r $w $lib: $fps frames/s$extra"
      echo "$w $lib $fps" >> $TMP
    done
  done
done
python - "$TMP" "$A" "$B" <<'PY'
import sys, collections
rows = collections.defaultdict(list)
for l in open(sys.argv[1]):
    w, lib, fps = l.split(); rows[(w, lib)].append(float(fps))
for w in ("sphere", "static"):
    a, b = rows[(w, sys.argv[2])], rows[(w, sys.argv[3])]
    if sys.argv[2] == sys.argv[3]:  # A against A: the pairs of one repetition in running order (what the order alone does)
        a, b = a[0::2], a[1::2]
    if a and b:
        print("%s: mean %.0f -> %.0f frames/s, B / A = %.4f (per repetition: %s)" % (w, sum(a) / len(a), sum(b) / len(b), (sum(b) / len(b)) / (sum(a) / len(a)),
              " ".join("%.4f" % (y / x) for x, y in zip(a, b))))
PY
rm -f $TMP
