#!/bin/bash
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
timeout -k 10 900 python -m pytest tests -m gpu -v -x > gpurun_out/pytest_gpu.log 2>&1; rc=$?
tail -4 gpurun_out/pytest_gpu.log; [ $rc -eq 0 ] || { tail -40 gpurun_out/pytest_gpu.log; exit 1; }
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 || exit 1
timeout -k 10 700 python bench.py --steps 20 --warmup 5 > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"
python3 -c "
import json; d=json.load(open('gpurun_out/bench.json')); print('images/s', round(d['value'],1), 'sustained', round(d['sustained']['images_per_s'],1), 'frac', round(d['model_mfma_frac'],3), 'check', d['output_check']); q=d['query']; print('batched', round(q['batched_qps']), 'single', round(q['single_query_qps']), 'c-abi', round(q['single_query_c_abi_qps']), 'fsd ms', q['find_similar_documents_ms']['median'], 'd2v', round(q['d2v_infer_docs_per_s']), 'train', round(q['d2v_train_doc_epochs_per_s'])); print('eva', d['eva02_large'].get('images_per_s_batch10'), 'ccip', d['ccip'].get('images_per_s_batch20'), d['ccip'].get('rerank_queries_per_s_single'))"
