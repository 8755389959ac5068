"""Import-time stand-in for torch_geometric, used ONLY by oracle/make_golden.py in the build
container so that /root/reference/main.py can be imported (PyG is not installed in this image).
It re-exports the oracle's restatement of the PyG operators -> results through it are
"[PyG, parity unpinned]".  Test infrastructure; never imported by gmlm_amd."""
