import faulthandler, sys, runpy
faulthandler.dump_traceback_later(100, exit=True)
sys.argv = ["bench.py"] + sys.argv[1:]
runpy.run_path("bench.py", run_name="__main__")
