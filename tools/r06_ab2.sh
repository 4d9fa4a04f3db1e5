cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
NREP=3 bash tools/hs_old_new.sh ISCA_FINISH_KERNEL=1
echo "---- moist: A (before T_v / lw_down changes) vs old (HEAD)"
for rep in 1 2; do for lib in A old; do echo "== $lib"; ISCA_DYN_LIB=$PWD/isca_amd/lib/libisca_dyn_$lib.so timeout 300 python tools/dev/moist_ab.py 2>&1 | tail -2 | cut -c1-150; done; done
