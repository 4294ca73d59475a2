#!/bin/bash
# Bisect kit, container side: the trees of earlier commits with their native libraries built, under bisect/<sha>/ (git-ignored, NOT
# gpurun-ignored: they travel to the GPU box with the snapshot), so that ONE lease can run suite + bench on the last GPU-verified commit,
# on intermediate ones and on HEAD, and attribute a regression of the chain kernels rewritten without a GPU in round 5.
#   tools/bisect_kit.sh                      1e71e26 (last commit a GPU ran, round 3) and 405ad63 (end of round 4)
#   tools/bisect_kit.sh <sha> ...            other commits
# GPU side: tools/gpu_settle.sh runs every bisect/*/ it finds (suite + `bench.py --no-cpu-baseline`) after HEAD's own pass.
set -eu
cd "$(dirname "$0")/.."
SHAS=${*:-"1e71e26 405ad63"}
mkdir -p bisect
for s in $SHAS; do
  d=bisect/$s
  if [ ! -f $d/sos_slam_amd/csrc/libsos_host.so ]; then
    rm -rf $d; mkdir -p $d
    git archive $s | tar -x -C $d
    rm -rf $d/profiles $d/DESIGN_APPENDIX.md $d/gpurun_out      # (evidence of past rounds: not needed to run the tree)
    (cd $d && python -c "import __graft_entry__ as g; g.build()" > build.log 2>&1) || { echo "build of $s failed: $d/build.log"; exit 1; }
  fi
  echo "$s: $(git log -1 --format=%s $s | cut -c1-100)"
  ls -la $d/sos_slam_amd/csrc/*.so | awk '{print "   ", $5, $9}'
done
# HEAD + the device-code rewrites no GPU has run (out of the default path since round 6, kept as patches): the batched-load forms of the
# four accumulate / stitch kernels (round 5) and the tracker's res_pixel written as selects (round 3, removes 16 B of scratch).
# bisect/pending -- the A/B of the first lease decides, patch by patch, what is applied
d=bisect/pending
rm -rf $d; mkdir -p $d
git archive HEAD | tar -x -C $d
rm -rf $d/DESIGN_APPENDIX.md $d/gpurun_out
(cd $d && patch -p1 -s < profiles/r06_chain_batched_loads.patch && patch -p1 -s < profiles/r06_tracker_res_pixel_selects.patch && rm -rf profiles && python -c "import __graft_entry__ as g; g.build()" > build.log 2>&1) || { echo "build of HEAD + pending patches failed: $d/build.log"; exit 1; }
echo "pending: HEAD $(git rev-parse --short HEAD) + profiles/r06_chain_batched_loads.patch + profiles/r06_tracker_res_pixel_selects.patch"
du -sh bisect
