set -e
export PSA_AMD_LIB=$PWD/parallel-sequence-alignment_amd/psa_amd/libpsa_debug.so
O=gpurun_out/stream_ab_r5b.txt
: > $O
PSA_STREAM_MIN=100000000 timeout -k 10 120 python tools/stream_ab.py >> $O 2>&1
timeout -k 10 120 python tools/stream_ab.py >> $O 2>&1
PSA_STREAM_CHUNK=262144 timeout -k 10 120 python tools/stream_ab.py >> $O 2>&1
PSA_STREAM_CHUNK=1048576 timeout -k 10 120 python tools/stream_ab.py >> $O 2>&1
PSA_STREAM_MIN=100000000 timeout -k 10 120 python tools/stream_ab.py >> $O 2>&1
PSA_STREAM_TRACE=1 timeout -k 10 120 python tools/stream_ab.py 2>&1 | tail -8 >> $O
cat $O
