#!/bin/bash
O=/root/repo/gpurun_out/r03q
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for args in "many 50" "many 50"; do
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o run -- python /root/repo/scripts/prof_repro.py $args > $O/repro.log 2>&1; rc=$?
  echo "== $args: rc $rc; last progress: $(grep '^run' $O/repro.log | tail -1)"
done
timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/pp2 -o run -- python /root/repo/scripts/prof_repro.py many 50 > $O/repro2.log 2>&1; echo "== no --stats: rc $?; last progress: $(grep '^run' $O/repro2.log | tail -1)"
GLX_CXXFLAGS="-DGLX_LOOP_FORM=0" python -m graphlearning_amd._build > /dev/null 2>&1
GLX_CXXFLAGS="-DGLX_LOOP_FORM=0" python -m graphlearning_amd._build > /dev/null 2>&1; cd /tmp
GLX_CXXFLAGS="-DGLX_LOOP_FORM=0" timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp3 -o run -- python /root/repo/scripts/prof_repro.py many 50 > $O/repro3.log 2>&1; echo "== loop form 0 build: rc $?; last progress: $(grep '^run' $O/repro3.log | tail -1)"
