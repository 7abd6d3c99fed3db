"""GPU box: bias_probe-style sweep timings with the in-tree library or an A/B build at scripts/_cut/libvocr.so (argv[1] == 'cut')."""
import os, sys, runpy
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import vistaocr_amd._lib as L
if len(sys.argv) > 1 and sys.argv[1] == "cut":
    L.LIB_PATH = os.path.join(root, "scripts", "_cut", "libvocr.so")
print("library:", L.LIB_PATH)
sys.argv = [sys.argv[0]]
runpy.run_path(os.path.join(root, "scripts", sys.argv[0] and "bias_probe.py"), run_name="__main__")
