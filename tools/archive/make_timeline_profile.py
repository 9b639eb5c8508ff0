import subprocess
h=subprocess.check_output(['git','log','--oneline','-1'],cwd='/root/repo').decode()[:7]
t=open('/root/repo/gpurun_out/trace_b1.txt').read().strip()
open('/root/repo/profiles/r03_single_frame_timeline.md','w').write('''# One 640x480 frame per call: the launch timeline (round 3, commit %s)

`tools/trace_b1.sh` on the GPU box (1x MI355X): `rocprofv3 --kernel-trace` of `python3 bench.py --steps 50 --warmup 5 --batch 1 --no-cpu-baseline
--no-extras` (ORBX_SPLIT=0), one call out of the middle of the timed region, times relative to its first kernel.  Round 2's timeline for the same
call: `k_pyr_chain` 19 + `k_fast` 11 + `k_octree` 33 + `k_describe` 6 = 65 us.

```
%s
```

Inside `k_pyr_cols` (a `-DORBX_CHAIN_STAMPS` build, `tools/cols_stamps.py`, region 21 of the 192 of the 40-px cut; `s_memrealtime`, 10-ns ticks):

```
loads + records + image rectangle in LDS                    0.96 us
level 1 derived, level 0 written                            1.20 us
level 2 derived, level 1 written                            0.84 us
level 3 derived, level 2 written                            0.76 us
level 4 derived, level 3 written                            0.68 us
level 5 derived, level 4 written                            0.68 us
level 6 derived, level 5 written                            0.76 us
level 7 derived, level 6 written                            0.68 us
level 7 written                                             0.44 us
total 7.00 us
192 workgroups: starts 0..1.20 us, duration mean 7.18 max 9.88 (the corner regions), last end 10.93 us
```

Inside `k_fast_wide` (`-DORBX_FAST_CLOCK` build, `tools/fast_spans.py`): a level-0 cell's workgroup — staged 1.12, score pass 0.93, NMS + count 1.03, emit 1.10 = 4.2 us;
3732 waves start within 0.4 us, the last one ends at 8.1 us (the last-dispatched workgroups run at a third of the first ones' speed).  Inside `k_describe`
(`-DORBX_DESC_STAMPS`, `tools/desc_spans.py`): weight table + per-level counters 1.52, level geometry + selection entry 0.12, patch loads 1.37, IC_Angle + rBRIEF 1.29:
4.9 us before the final stores.

Inside `k_octree_256r`, the level-0 workgroup (`-DORBX_OCT_STAMPS` build, `tools/oct_stamps.py`): leaf tables + roots 1.44, count pyramid 1.00 + 0.32, fast-forward
1.76, the one sorted pass (64 -> 217 nodes, on one wave) 2.24, best keys + output 1.68: 8.4 us; the coarsest level needs two sorted passes and ends last, at 9.7 us.
''' % (h, t))
