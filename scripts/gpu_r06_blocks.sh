#!/bin/bash
# Round 6, VERDICT r05 item 4: row blocks of the scale-24 graph (scrambled labels) with and without the set-up step that hands the library the
# global column counts (GrX_Matrix_shard_setup), the Kronecker-26 block, and the N-rank code path with one rank over RCCL.
OUT=gpurun_out/r06j; mkdir -p $OUT; export TMPDIR=/tmp
[ -n "$VARIANT" ] && export GRB_MI355X_LIB=$PWD/build/variants/$VARIANT/libgrb_mi355x.so
python -m pytest tests/test_vertex_order.py -m gpu -x -q -k "shard_setup" 2>&1 | tail -2
blk() { tag=$1; shift; timeout 900 python bench.py "$@" --no-cpu-baseline --no-extra 2>$OUT/$tag.err | tee $OUT/$tag.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],4), 'frac', round(d['roofline']['frac'],3), 'verified', d['verified'], 'ordered', d['stats'].get('ordered'), 'reorders', d['stats'].get('reorders'), 'build_ms', round(d['layout_build_call_ms'],1), 'setup', d.get('shard_setup'))"; }
for w in 2 4 8; do blk block0of${w}_setup --block 0/$w; blk block0of${w}_nosetup --block 0/$w --no-shard-setup; done
blk block7of8_setup --block 7/8
blk kron26_block0of8_setup --workload kron26 --block 0/8
blk kron26_block0of8_nosetup --workload kron26 --block 0/8 --no-shard-setup
blk bfsstep_block0of8_setup --workload mxv_lor_land_masked --block 0/8
blk bfsstep_block0of8_nosetup --workload mxv_lor_land_masked --block 0/8 --no-shard-setup
blk force_dist --force-dist
blk force_dist_chunks2 --force-dist --overlap-chunks 2
