import sys, time
sys.path.insert(0, "/root/repo")
from oracle import oracle as O
O.build()
from genlib_jl_amd import synth
ind, fa, mo, sex, pro = synth.random_mating(100_000, 10_000, 20, skip_permille=50)
op = O.Pedigree(ind, fa, mo)
print("default threads", O.num_threads(), "usable", O.usable_cpus())
t = time.time(); op.phi(pro, stop_after_levels=6); print("default: %.2f s" % (time.time() - t))
print("fitted", O.fit_threads_to_quota())
t = time.time(); op.phi(pro, stop_after_levels=6); print("fitted: %.2f s" % (time.time() - t))
