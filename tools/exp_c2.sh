set -e -o pipefail
cd $GRAFT_REPO_ROOT
run() { echo "$@"; env "$@" timeout -k 10 120 python tools/blocking.py C3 40 | tail -1; }
run A=1
run XRT_TUNE=64,16,48,32
run XRT_TUNE=64,8,48,32
run XRT_TUNE=64,32,48,32
run XRT_TUNE=64,16,24,32
run XRT_TUNE=64,16,96,32
run XRT_TUNE=64,16,48,16
run XRT_TUNE=64,16,48,48
run XRT_TUNE=48,16,48,32
run XRT_BATCH_MAX=32
run XRT_BATCH_MAX=128
run XRT_FIRST_BATCH=128
run XRT_NO_FEEDBACK=1
run XRT_LONG_FRAC=4,10
run XRT_LONG_FRAC=1,3
