cd $GRAFT_REPO_ROOT
T=$(python -c 'import torch,os;print(os.path.dirname(torch.__file__)+"/lib")')
echo "== system runtime"; ./scripts/probes/graph_memset_probe
echo "== the runtime bundled with PyTorch (LD_PRELOAD)"; LD_PRELOAD=$T/libamdhip64.so LD_LIBRARY_PATH=$T:$LD_LIBRARY_PATH ./scripts/probes/graph_memset_probe
