# region_diff at config 4: how far the search speculates (PLP_RDIFF_SPEC_*) against its time.   gpurun -- 'bash scripts/debug/rdiff_spec_sweep.sh'
run() { echo "== $*"; env "$@" PLP_RDIFF_STATS=1 python scripts/debug/rdiff_call_profile.py 2>&1 | grep -E "plp_region_diff_search|ms, " | tail -3 | sed -e 's/beyond 64 rows.*search loop/search loop/' -e 's/plp_region_diff_search: //' -e 's/ (.*requests//' | tr '\n' ' '; echo; }
for rep in 1 2; do
run A=1
for sib in 50 100 150 200 300 600; do run PLP_RDIFF_SPEC_SIB=$sib PLP_RDIFF_SPEC_DEEP=0; done
run PLP_RDIFF_SPEC_SIB=100 PLP_RDIFF_SPEC_DEEP=0 PLP_RDIFF_SPEC_CHAIN_SCAN=10 PLP_RDIFF_SPEC_CHAIN_NODE=12
run PLP_RDIFF_SPEC_SIB=100 PLP_RDIFF_SPEC_DEEP=0 PLP_RDIFF_SPEC_CHILD=0
done
