bash tools/collect_profiles_r05.sh > gpurun_out/collect5.log 2>&1; tail -5 gpurun_out/collect5.log
