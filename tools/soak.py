"""A short REAL training run on the vectorised path (hip_graph runner, in-place replay, graph-captured train step): does the loop
learn?  Prints, every --every iterations, the mean collective return of the training rollouts (with exploration), of a greedy test
rollout on all envs, epsilon, the learner's losses and the size of the parameters; checks that everything stays finite.

    python tools/soak.py [--config cleanup5] [--n-env 4096] [--iters 1500] [--train-steps-per-rollout 8] [--every 100]

Schedules (run.py setup): with batch_size_run > 1 the epsilon clock advances by episode_limit per ROLLOUT and the target network is
synchronised every target_update_interval learner.train calls (schedule_unit "rollouts"); `--schedule-unit env_steps` shows the
reference's literal arithmetic at 4096 envs (target sync after every step, epsilon at its floor after one rollout)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch as th  # noqa: E402

from bench import CONFIGS  # noqa: E402
from homophily_marl_amd.run import load_config, setup, train_iteration  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="cleanup5", choices=sorted(CONFIGS))
ap.add_argument("--n-env", type=int, default=4096)
ap.add_argument("--iters", type=int, default=1500)
ap.add_argument("--train-steps-per-rollout", type=int, default=8)
ap.add_argument("--every", type=int, default=100)
ap.add_argument("--schedule-unit", default=None)
ap.add_argument("--epsilon-anneal-time", type=int, default=None)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--anomaly", action="store_true", help="torch.autograd.set_detect_anomaly(True)")
ap.add_argument("--nan-check", action="store_true", help="after every iteration: first non-finite gradient / parameter, with details")
ap.add_argument("--set", action="append", default=[], help="extra config override key=value (python literal)")
a = ap.parse_args()
c = CONFIGS[a.config]
N, n, T = a.n_env, c["n_agents"], 100
over = dict(runner="hip_graph", train_graph=1, batch_size_run=N, batch_size=16, buffer_size=-(-5000 // N) * N, buffer_cpu_only=False,
            store_state=False, train_steps_per_rollout=a.train_steps_per_rollout,
            env_args=dict(num_agents=n, map=c["map"], episode_limit=T, view_size=c["view_size"], seed=a.seed),
            use_cuda=True, save_model=False, runner_stats=False, learner_log_interval=10 ** 12)
import ast
for kv in a.set:
    k, v = kv.split("=", 1)
    over[k] = ast.literal_eval(v)
if a.schedule_unit:
    over["schedule_unit"] = a.schedule_unit
if a.epsilon_anneal_time:
    over["epsilon_anneal_time"] = a.epsilon_anneal_time
cfg = load_config(c["env"], overrides=over)
th.manual_seed(a.seed)
if a.anomaly:
    th.autograd.set_detect_anomaly(True)
ctx = setup(cfg)
print("config %s, %d envs, %d learner.train per rollout, schedule_unit %s, epsilon %g -> %g over %d clock steps, target sync every %d" % (
    a.config, N, a.train_steps_per_rollout, ctx.args.schedule_unit, ctx.args.epsilon_start, ctx.args.epsilon_finish,
    ctx.args.epsilon_anneal_time, ctx.args.target_update_interval), flush=True)
ep, t0 = 0, time.time()
acc, cnt = 0.0, 0
if a.nan_check:      # check after EVERY learner.train: the first step whose gradient is not finite, and where its forward breaks
    _train = ctx.learner.train
    _prev = {}

    def checked(batch, t_env, episode):
        before = {k: v.clone() for k, v in ctx.mac.agent.state_dict().items()}
        _train(batch, t_env, episode)
        g = ctx.learner._flat_grad
        if g is not None and not bool(th.isfinite(g).all()):
            print("FIRST NON-FINITE GRADIENT at train step", ctx.train_steps, "params before the step finite:",
                  all(bool(th.isfinite(v).all()) for v in before.values()), "max |param| before", max(float(v.abs().max()) for v in before.values()))
            ctx.mac.agent.load_state_dict(before)
            sb = ctx.learner._static_batch if ctx.learner._graph is not None else batch
            with th.no_grad():
                mac = ctx.mac
                sh = mac.unroll_shared(sb)
                B, T, n = sb.batch_size, sb.max_seq_length, mac.n_agents
                parts, wh, bh = mac.unroll_pre(sb, sh)              # [gi_env, gi_inc] (each [n, T * B, 192]), weight / bias parts
                for nm, gi in zip(("gi_env", "gi_inc"), parts):
                    print("   %s finite" % nm, bool(th.isfinite(gi).all()), "max", float(gi[th.isfinite(gi)].abs().max()))
                print("   wh max", max(float(w.abs().max()) for w in wh), " bh max", max(float(x.abs().max()) for x in bh))
                from homophily_marl_amd import ops
                for nm, hs in zip(("h_env", "h_inc"), ops.gru_sequence_parts(parts, T, B, wh, bh)):
                    bad = (~th.isfinite(hs)).nonzero()
                    print("   %s finite" % nm, bad.numel() == 0, " first non-finite index:", bad[0].tolist() if bad.numel() else None, " count", bad.shape[0])
                L = ctx.learner
                tgt_ok = all(bool(th.isfinite(v).all()) for v in L.target_mac.agent.state_dict().values())
                print("   target net finite", tgt_ok, " optimiser state finite",
                      all(bool(th.isfinite(st[k]).all()) for o in (L.optimiser_env, L.optimiser_inc) for st in o.state.values() for k in ("exp_avg", "exp_avg_sq", "step")))
                q_env, q_inc, tq_env, tq_inc = L.unroll_pair(sb)
                for nm, t in (("q_env", q_env), ("q_inc", q_inc), ("tq_env", tq_env), ("tq_inc", tq_inc)):
                    print("   eager", nm, "finite", bool(th.isfinite(t).all()), "max", float(t[th.isfinite(t)].abs().max()))
                d = L.denominators(sb)
                print("   dens eager", d.tolist(), "static", L._static_dens.tolist())
            if L._graph is not None:
                L._graph[0].replay(); th.cuda.synchronize()
                print("   graph replay on the restored weights: grad finite", bool(th.isfinite(L._flat_grad).all()),
                      {k: float(v) for k, v in L._static_logs.items() if k.startswith("loss") or k.startswith("q_")})
            # the same step eagerly (no graph): where does the backward break?
            L._flat_grad.zero_()
            logs = L.forward_backward(sb, L._static_dens)
            print("   eager forward_backward (fused loss): grad finite", bool(th.isfinite(L._flat_grad).all()), "loss_inc", float(logs["loss_value_inc"]))
            off = 0
            for name, prm in ctx.mac.agent.named_parameters():
                sl = L._flat_grad[off:off + prm.numel()]; off += prm.numel()
                print("      %-22s finite %s  max %.3e" % (name, bool(th.isfinite(sl).all()), float(sl[th.isfinite(sl)].abs().max()) if bool(th.isfinite(sl).any()) else float("nan")))
            L.args.fused_loss = False
            L._flat_grad.zero_()
            logs = L.forward_backward(sb, L._static_dens)
            print("   eager forward_backward (tensor-op loss): grad finite", bool(th.isfinite(L._flat_grad).all()), "loss_inc", float(logs["loss_value_inc"]))
            th.save({k: v.cpu() for k, v in sb.data.transition_data.items()}, "/root/repo/gpurun_out/r02e/nan_batch.pt")
            th.save({k: v.cpu() for k, v in before.items()}, "/root/repo/gpurun_out/r02e/nan_weights.pt")
            th.save({k: v.cpu() for k, v in L.target_mac.agent.state_dict().items()}, "/root/repo/gpurun_out/r02e/nan_target.pt")
            raise SystemExit(1)
    ctx.learner.train = checked
for it in range(a.iters):
    ep = train_iteration(ctx, ep)
    acc += float(ctx.runner.ep_return.sum(-1).mean()); cnt += 1
    if a.nan_check:
        g = ctx.learner._flat_grad
        bad_g = g is not None and not bool(th.isfinite(g).all())
        bad_p = [k for k, v in ctx.mac.agent.state_dict().items() if not bool(th.isfinite(v).all())]
        if bad_g or bad_p:
            logs = {k: float(v) for k, v in ctx.learner._static_logs.items()}
            print("NON-FINITE at iteration %d: grad finite %s (|g| max %s), bad params %s" % (it + 1, not bad_g, float(g.abs().max()), bad_p[:6]))
            print("   logs", logs)
            print("   dens", ctx.learner._static_dens.tolist())
            off = 0
            for name, prm in ctx.mac.agent.named_parameters():
                sl = g[off:off + prm.numel()]; off += prm.numel()
                if not bool(th.isfinite(sl).all()):
                    print("   non-finite grad in", name, int((~th.isfinite(sl)).sum()), "of", sl.numel())
            for opt, nm in ((ctx.learner.optimiser_env, "env"), (ctx.learner.optimiser_inc, "inc")):
                mx = max(float(st["exp_avg_sq"].max()) for st in opt.state.values())
                print("   Adam", nm, "max exp_avg_sq", mx, "step", float(next(iter(opt.state.values()))["step"]))
            sb = ctx.learner._static_batch
            print("   batch reward range", float(sb["reward"].min()), float(sb["reward"].max()), "actions range", int(sb["actions"].min()), int(sb["actions"].max()),
                  "actions_inc range", int(sb["actions_inc"].min()), int(sb["actions_inc"].max()), "obs finite", bool(th.isfinite(sb["obs"].float()).all()),
                  "pos range", float(sb["agent_pos"].min()), float(sb["agent_pos"].max()), "apple_den", float(sb["apple_den"].min()), float(sb["apple_den"].max()))
            break
    if it % a.every == a.every - 1 or it == a.iters - 1:
        th.cuda.synchronize()
        logs = {k: float(v) for k, v in ctx.learner._static_logs.items()} if getattr(ctx.learner, "_static_logs", None) else {}
        eps = ctx.mac.action_selector.epsilon
        ctx.runner.run(test_mode=True)
        greedy = float(ctx.runner.ep_return.sum(-1).mean())
        p = th.cat([q.detach().flatten() for q in ctx.mac.parameters()])
        print("iter %4d  env-steps %10d  train steps %6d  eps %.3f  collective return: train %6.2f (mean of last %d rollouts)  greedy %6.2f  "
              "loss_env %.5f loss_inc %.5f loss_sim %.4f  q_env %.4f q_inc %.4f  |grad| %.3f  clean/step %.4f  |theta| %.2f  finite %s  %.0fs" % (
                  it + 1, ctx.runner.t_env, ctx.train_steps, eps, acc / cnt, cnt, greedy, logs.get("loss_value_env", float("nan")),
                  logs.get("loss_value_inc", float("nan")), logs.get("loss_sim", float("nan")), logs.get("q_env_taken_mean", float("nan")),
                  logs.get("q_inc_taken_mean", float("nan")), float(ctx.learner._flat_grad.norm()),
                  float(ctx.runner.store["clean_num"][:, :-1].mean()), float(p.norm()), bool(th.isfinite(p).all()), time.time() - t0), flush=True)
        acc, cnt = 0.0, 0
        assert bool(th.isfinite(p).all())
assert ctx.runner.env.native.poll_error() == 0
