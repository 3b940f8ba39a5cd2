for i in 1 2 3; do for v in base bw5 bw7; do
if [ $v = base ]; then L=""; else L="EVC_LIB=/root/repo/scripts/libevc_$v.so"; fi
echo "== $v $(env $L timeout 300 python scripts/lstm_layer_bench.py --shape teacher 2>&1 | grep "bwd" | sed 's/.*| fwd/fwd/')"
done; done
for v in base bw5 bw7; do
if [ $v = base ]; then L=""; else L="EVC_LIB=/root/repo/scripts/libevc_$v.so"; fi
echo "== $v $(env $L timeout 300 python scripts/lstm_layer_bench.py --shape student 2>&1 | grep "bwd" | sed 's/.*| fwd/fwd/')"
done
