cd $GRAFT_REPO_ROOT
timeout 120 tools/bin/chain_probe
