cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for cfg in "1 2" "2 2" "3 2" "4 2" "1 1" "3 1"; do set -- $cfg
for i in 1 2; do
ES_ROWS_CAV_SLICES=$1 ES_ROWS_VO1_SLICES=$2 timeout 600 python bench.py --workload layout --steps 300 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c100-150 | sed "s/^/cav_slices=$1 vo1_slices=$2 /"
done; done
ES_ROWS_CAV_SLICES=3 timeout 900 python -m pytest tests/test_hip_rows.py -x -q -k "unet1d or layout" 2>&1 | tail -2
