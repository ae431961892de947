#!/bin/bash
# round 3 final profiles: headline, curvature, the slab shape, the slab-path timeline
bash scratch/prof.sh $1 > /dev/null || exit 1
bash scratch/prof.sh $1_curv --reg curvature > /dev/null || exit 1
bash scratch/prof.sh $1_slab --cols 512 > /dev/null || exit 1
bash scratch/slab_timeline.sh $1 > gpurun_out/slab_timeline_$1.txt 2>&1 || { tail gpurun_out/slab_timeline_$1.txt; exit 1; }
tail -5 gpurun_out/prof_$1/summary.txt
