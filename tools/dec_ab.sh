# developer A/B of the decode kernel by knob: split counts (SGLK_DEC_SPLITS), cache-write fold (SGLK_DEC_FOLD), row policy (SGLK_DEC_NT)
# usage: bash tools/dec_ab.sh "<shape> ..." "<ENV=val ENV=val> ..."     (a configuration is one word: join several settings with commas)
cd $GRAFT_REPO_ROOT
SHAPES=${1:-"128,22,1,576,512,4096,1 64,32,4,128,128,4096,0 40,22,1,576,512,1064,1"}
CONFIGS=${2:-"default SGLK_DEC_FOLD=0 SGLK_DEC_NT=0"}
for rep in 1 2; do
for shape in $SHAPES; do
for cfg in $CONFIGS; do
    if [ $cfg = default ]; then envs=""; else envs=$(echo $cfg | tr ',' ' '); fi
    echo "$cfg $(env $envs SGLK_DEC_SHAPE=$shape timeout -k 10 120 python tools/decode_probe.py 2>/dev/null | tail -1)"
done
done
done
