# developer A/B: decode-size expert weight streams with the default / non-temporal load policy (SGLK_W_NT=0 / 1 / unset = the rule)
# usage: bash tools/nt_ab.sh [fp8|int8]
cd $GRAFT_REPO_ROOT
KIND=${1:-fp8}
for rot in 1 3; do
for shape in 2048,768,128,8 7168,384,256,8; do
for m in ${NT_AB_M:-1 4 8 16 64 256 1024}; do
    r=$rot; [ $rot = 3 ] && [ $shape = 7168,384,256,8 ] && r=2
    for nt in 0 1 auto; do
        v=$nt; [ $nt = auto ] && v=
        echo "nt=$nt $(SGLK_W_NT=$v SGLK_PROBE_ROTATE=$r SGLK_PROBE_SHAPE=$shape timeout -k 10 120 python tools/stage_probe.py $KIND $m 2>/dev/null | tail -1)"
    done
done
done
done
