# A/B of two builds of tools/experiments/resident_probe.hip (the resident SART sweep alone): ab_probe.sh <probe A> <probe B>
cd $GRAFT_REPO_ROOT/tools/experiments
for i in 1 2 3; do
./$1 512 90 64 1 5 2>&1 | tail -1
./$2 512 90 64 1 5 2>&1 | tail -1
done
./$2 256 60 256 1 3 2>&1 | tail -1
./$2 512 90 64 1 3 1 2>&1 | tail -1
./$2 96 12 130 2 3 2>&1 | tail -1
