bash tools/collect_profiles_r05.sh b > gpurun_out/collect5b.log 2>&1; rc=$?; tail -5 gpurun_out/collect5b.log; exit $rc
