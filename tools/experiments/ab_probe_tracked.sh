# A/B of two probe builds on the TRACKED sweep (last argument 1): ab_probe_tracked.sh <probe A> <probe B>
cd $GRAFT_REPO_ROOT/tools/experiments
for i in 1 2 3; do
./$1 512 90 64 1 5 1 2>&1 | tail -1
./$2 512 90 64 1 5 1 2>&1 | tail -1
done
./$1 512 90 512 1 3 1 2>&1 | tail -1
./$2 512 90 512 1 3 1 2>&1 | tail -2
./$2 256 60 256 1 3 1 2>&1 | tail -1
./$2 96 12 130 2 3 1 2>&1 | tail -2
./$2 512 90 64 1 3 2>&1 | tail -1
