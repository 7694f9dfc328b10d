L=conv2_1,conv2_2,conv3_2,conv4_2,conv5_2
for i in 1 2; do
echo NEW; ODTK_DBG2=1 python tools/conv_bench.py $L fwd,dgrad 30 0 2>&1 | grep "conv\|sum"
echo OLD; ODTK_LIB=tools/probes/bin/libodtk_old.so python tools/conv_bench.py $L fwd,dgrad 30 0 2>&1 | grep "conv\|sum"
done
