#!/bin/bash
# Round 3: full GPU suite, then the profile refresh.
set -e -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03g
rm -rf $O && mkdir -p $O
timeout -k 10 1500 python3 -m pytest tests -q -s -m gpu > $O/tests.log 2>&1 && TRC=0 || TRC=$?
grep -E "^(FAILED|ERROR)|passed|failed" $O/tests.log | tail -15 || true
[ $TRC -eq 0 ] || grep -n "^E " $O/tests.log | head -30
echo "tests rc=$TRC"
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 && echo "smoke ok" || { tail -5 $O/smoke.log; }
bash tools/refresh_profiles.sh > $O/refresh.log 2>&1 || { tail -20 $O/refresh.log; exit 1; }
tail -5 $O/refresh.log
