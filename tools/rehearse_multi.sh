# rehearsals of the N > 1 control flow on a one-GPU box (ranks share the GPU, gloo): launcher paths and the config-4 shards
mkdir -p gpurun_out/r2o
export GAAST_BENCH_REHEARSAL=1
echo "--- self-launch, 4 ranks, ragged-free small batch"
python bench.py --gpus 4 --batch 1000 --steps 2 --warmup 1 > gpurun_out/r2o/self4.json 2> gpurun_out/r2o/self4.err; echo rc=$?
python -c "
import json; d=json.load(open('gpurun_out/r2o/self4.json')); print(d['n_gpus'], d['global_batch'], d['config']['shards'], d['rccl_ranks'], d['scaling'], d['gather']['chunks'], 'value', '%.3g' % d['value'])"
echo "--- torch.distributed.run, 2 ranks, BASELINE configs[3] shards (524288 items per rank on ONE GPU)"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 > gpurun_out/r2o/torchrun2.json 2> gpurun_out/r2o/torchrun2.err; echo rc=$?
python -c "
import json; d=json.load(open('gpurun_out/r2o/torchrun2.json')); print(d['n_gpus'], d['global_batch'], d['config']['shards'], d['config']['workload'], d['scaling'], d['gather'].get('error'), 'value', '%.3g' % d['value'], 'ms', d['ms_per_step'])"
tail -3 gpurun_out/r2o/torchrun2.err
