# developer A/B of the decode kernel: libsglk_old.so (previous loop, if built) against the product library and its knobs
cd $GRAFT_REPO_ROOT
OLD=$GRAFT_REPO_ROOT/sgl-cpu-tests_amd/sgl_kernel/libsglk_old.so
for rep in 1 2; do
for shape in 40,22,1,576,512,1064,1 1,22,1,576,512,1024,1 16,32,8,128,128,2048,0 8,32,4,128,128,8192,0 64,32,4,128,128,4096,0; do
    [ -f $OLD ] && echo "old     $(SGLK_LIB_PATH=$OLD SGLK_DEC_SHAPE=$shape timeout -k 10 120 python tools/decode_probe.py 2>/dev/null | tail -1)"
    echo "nofold  $(SGLK_DEC_FOLD=0 SGLK_DEC_SHAPE=$shape timeout -k 10 120 python tools/decode_probe.py 2>/dev/null | tail -1)"
    echo "new     $(SGLK_DEC_SHAPE=$shape timeout -k 10 120 python tools/decode_probe.py 2>/dev/null | tail -1)"
done
done
