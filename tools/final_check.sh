O=gpurun_out/final_check
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/tests.log 2>&1; rc=$?; echo "rc=$rc" >> $O/tests.log; tail -5 $O/tests.log
[ $rc -eq 0 ] || exit 1
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -3 $O/smoke.txt
timeout -k 10 900 bash tools/run_profile_r03.sh > $O/profile.log 2>&1; echo "profile rc=$?"; tail -5 $O/profile.log
