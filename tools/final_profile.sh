#!/bin/bash
# End-of-round measurement set on the GPU box -> gpurun_out/final/ (copy what is judged into profiles/).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; rm -rf $O; mkdir -p $O
cd $R && python -m pytest tests -q -m gpu 2>&1 | tail -3 > $O/pytest_gpu.txt; cd /tmp
python3 $R/bench.py > $O/bench.json 2> $O/bench.err
python3 $R/bench.py --workload C2 --no-cpu-baseline > $O/bench_c2.json 2>> $O/bench.err
python3 $R/bench.py --workload C5 --no-cpu-baseline > $O/bench_c5.json 2>> $O/bench.err
VSP_GENERATOR=f16 python3 $R/bench.py --no-cpu-baseline > $O/bench_f16mode.json 2>> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python3 $R/bench.py --no-cpu-baseline > $O/bench_traced.json 2>> $O/bench.err
python3 $R/tools/trace_fused.py $O/trace/t_kernel_trace.csv > $O/generator_per_launch.txt
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>> $O/bench.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>> $O/bench.err
python3 $R/tools/traffic_from_pmc.py 'cl_conv_f16s|cl_respair_f16s' $O/pmc_fetch/p_counter_collection.csv $O/pmc_write/p_counter_collection.csv f16s 64 > $O/traffic.json
cat $O/pytest_gpu.txt; cut -c1-400 $O/bench.json; tail -3 $O/generator_per_launch.txt; cat $O/traffic.json
