"""Diagnostic: kernel mix of learner.train (graph-captured).  rocprofv3 --kernel-trace --stats -- python3 tools/train_prof.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch as th
from homophily_marl_amd.run import load_config, setup
N = int(os.environ.get("N_ENV", 512))
cfg = load_config("cleanup", overrides=dict(runner="hip_graph", train_graph=int(os.environ.get("TRAIN_GRAPH", 1)), batch_size_run=N, batch_size=16, buffer_size=N,
                                             buffer_cpu_only=False, store_state=False, obs_storage=os.environ.get("OBS_STORAGE", "code"),
                                             env_args=dict(num_agents=5, map="default5", episode_limit=100, seed=1),
                                             use_cuda=True, save_model=False, runner_stats=False, learner_log_interval=10 ** 12))
th.manual_seed(0)
ctx = setup(cfg)
batch = ctx.runner.run(False)
ctx.buffer.insert_episode_batch(batch)
# the loop body of run.train_iteration without the rollout: device-side sample straight into the captured step's batch, then train; the
# learner's episode counter is the number of train calls (schedule_unit "rollouts": target sync every 20 calls)
for i in range(int(os.environ.get("TRAIN_CALLS", 40))):
    if i == 10:
        th.cuda.synchronize(); t0 = time.perf_counter()
    sample = ctx.buffer.sample(16, out=ctx.learner.sample_out())
    ctx.learner.train(sample, 100 * N, i)
th.cuda.synchronize()
print("sample + train: %.2f ms/call" % (1e3 * (time.perf_counter() - t0) / (int(os.environ.get("TRAIN_CALLS", 40)) - 10)), flush=True)

if os.environ.get("TORCH_PROFILE"):
    # op-level view of ONE eager step: which aten ops the ~500 launches of a train step come from
    from torch.profiler import profile, ProfilerActivity
    ctx.learner.use_graph = False
    sample = ctx.buffer.sample(16)
    ctx.learner.cal_loss_and_step(sample)
    th.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        ctx.learner.cal_loss_and_step(sample)
        th.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=60, max_name_column_width=60))
    if os.environ.get("TORCH_PROFILE") == "shapes":
        print(prof.key_averages(group_by_input_shape=True).table(sort_by="cuda_time_total", row_limit=120, max_name_column_width=40, max_shapes_column_width=90))
