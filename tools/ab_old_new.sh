#!/bin/bash
# Same-box A/B of two BUILDS of the library (boxes differ by +-5 %, so two commits are compared by alternating their
# bench.py on ONE box): `build/old/` holds a second working tree with its own libisg_hip.so (git-ignored, but it travels to the
# GPU box), this tree is "new".  Prepare the old tree on the build host, e.g.
#   git worktree add -f /tmp/oldtree <commit> && (cd /tmp/oldtree && python -c "import __graft_entry__ as g; g.build()")
#   mkdir -p build/old && cp -r /tmp/oldtree/{intrinsic-subgraph-generation-for-vqa_amd,oracle,include,bench.py,__graft_entry__.py,BASELINE.json,tools,isubgvqa_amd.py} build/old/
# then   gpurun -- 'bash tools/ab_old_new.sh'   prints ms/step and the layer kernel's average launch (HIP events) per run.
for i in 1 2; do
  for v in old new; do
    if [ $v = old ]; then B=build/old/bench.py; else B=bench.py; fi
    timeout -k 10 200 python $B --no-cfg5 --steps 60 --warmup 10 > gpurun_out/ab_$v$i.json 2> gpurun_out/ab_$v$i.err
    python - <<PY
import json
d=json.loads(open("gpurun_out/ab_$v$i.json").read().strip().splitlines()[-1])
print("$v$i", d["ms_per_step"], d["roofline"]["avg_launch_us"])
PY
  done
done
