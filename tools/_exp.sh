R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_x5; mkdir -p $O; cd $R
for l in 1 2; do python tools/ab_builds.py run --mode none --passes 3 --lpt $l >> $O/pfpos.jsonl 2>&1; done
for l in 1 2; do python tools/ab_builds.py run --mode table --no-export --passes 2 --lpt $l >> $O/pfpos.jsonl 2>&1; done
cut -c1-220 $O/pfpos.jsonl
