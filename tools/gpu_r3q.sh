cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for v in 0 4 8 12 0; do echo "== ES_CONV_STAGGER=$v"; ES_CONV_STAGGER=$v timeout 600 python tools/conv_launch_table.py 2>&1 | grep "summed\|taps 27  Cin  224+0    N  224  @16x16x16 mode 0 epi 0 res . f32 1 f16 0\|taps 27  Cin  448+0    N  224  @16\|Cin 1120+0    N  448\|taps 27  Cin  224+448"; done
