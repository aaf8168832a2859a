cd $GRAFT_REPO_ROOT
for v in 84 164; do for sp in 2 3 4 6 12; do echo "TH=$v split=$sp"; WX_EMBED_SMALL_TH=$v WX_EMBED_SPLIT=$sp python tools/stage_classes.py C1 bf16 embed_patch 2>&1 | grep embed_patch; done; done
