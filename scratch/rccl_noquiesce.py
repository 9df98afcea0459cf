"""bench.py with GraphedTrainStep's pre-capture quiesce of the process-group watchdog disabled (the A side of the A/B that
motivated it): python scratch/rccl_noquiesce.py <bench.py arguments>"""
import os, runpy, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import vmrframe_amd.trainer as t
t.GraphedTrainStep._quiesce_process_group = staticmethod(lambda: None)
sys.argv = [os.path.join(root, "bench.py")] + sys.argv[1:]
runpy.run_path(os.path.join(root, "bench.py"), run_name="__main__")
