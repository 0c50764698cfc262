set -e
python tools/make_stream.py /dev/shm/t4k.webm 4k 60 6 
export LD_LIBRARY_PATH=$PWD/easyav1_amd:/opt/rocm/lib:$LD_LIBRARY_PATH
EASYAV1_AMD_TRACE=1 EASYAV1_AMD_GOPS=4 ./tools/bin/easyav1_bench /dev/shm/t4k.webm > gpurun_out/r03_f_trace_4k_gops4.txt 2>&1
EASYAV1_AMD_TRACE=1 EASYAV1_AMD_GOPS=1 ./tools/bin/easyav1_bench /dev/shm/t4k.webm > gpurun_out/r03_f_trace_4k_gops1.txt 2>&1
rm -f /dev/shm/t4k.webm
head -40 gpurun_out/r03_f_trace_4k_gops4.txt
