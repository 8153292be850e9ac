for rep in 1 2; do
for v in "GENPHI_D2H_SYM=0" "GENPHI_D2H_THREADS=12" "GENPHI_D2H_THREADS=15" "GENPHI_D2H_THREADS=24"; do echo "[$v]"; env $v python profiles/microbench/first_call_d2h.py cfg4 2>&1 | grep result_to_host | cut -c1-70; done; done
