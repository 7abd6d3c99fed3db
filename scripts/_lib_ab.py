"""GPU box: run another script of this directory with the in-tree library or with an A/B build at scripts/_cut/libvocr.so:
python scripts/_lib_ab.py [cut] <script.py> [its arguments]   (e.g. `cut ../bench.py --no-cpu-baseline`)"""
import os, sys, runpy
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import vistaocr_amd._lib as L
args = sys.argv[1:]
if args and args[0] == "cut":
    L.LIB_PATH = os.path.join(root, "scripts", "_cut", "libvocr.so")
    args = args[1:]
print("library:", L.LIB_PATH)
sys.argv = list(args)                   # the script and ITS arguments
runpy.run_path(os.path.join(root, "scripts", args[0]), run_name="__main__")
