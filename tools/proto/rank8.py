import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tools'))
import rank_compute_time as r
print(r.measure(8, 4, True, steps=200))
