"""
Measurement helpers of bench.py: the `roofline` object of the driver's JSON line from the engine's HIP-event profile
(Plan.profile / Plan.profile_aux) and from the committed rocprofv3 counter passes (profiles/r05_pmc_traffic.json).
Nothing here is on the routing path.
"""
from __future__ import annotations

import json
import os

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HBM_PEAK_GBS = 8000.0           # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
PMC_JSON = os.path.join(REPO, 'profiles', 'r05_pmc_traffic.json')

TILE_STATE_BYTES = 72.0      # per position and task: lag, first upstream, counts, ghost link (16), ss, sq, c1, c2, c3 read (40), sq, ss written (16)


def roofline_from_profile(prof: dict, nsub: int, peak_gbs: float = 8000.0, copy_gbs=None, unit: bool = False, kernel: str = None, traffic: dict = None):
    """`roofline` object of bench.py from Plan.profile() for the dominant kernel.

    Time-tiled kernel (k_tile; a launch advances its tiles by K ticks): ALGORITHMIC bytes of a launch = positions x
    (16 B x K: the record of every position read once and written once, a ghost's by the tile that owns its reach) +
    positions x 72 B (80 with the channel state of UnitMuskingum) of state and coefficients once per task; `frac` =
    those bytes / HIP-event time of the sampled launches / peak, a fraction of the roofline by construction.
    Every fourth launch is sampled, fill and drain launches included, so `avg_launch_us` is the average rocprofv3
    reports for the kernel.  Direct row path (k_direct; a launch routes K rows of every column): 16 B per reach-row (the lateral
    value read, the discharge written) + 64 B of per-column constants and state per task; every launch is sampled.
    `traffic` = HBM bytes per full launch measured by separate rocprofv3 --pmc passes, with the file it came from.
    `streaming_model_*` prices the same launches at SURVEY section 8(d)'s contract figure (72 B per reach sub-step + 16 B
    per reach row: what a kernel that keeps nothing on chip between ticks would move); a time-tiled kernel undercuts
    it, so that figure may exceed the peak and is NOT the roofline fraction."""
    if prof['sampled'] <= 0 or prof['sampled_ms'] <= 0:
        return None
    tpl = max(1, prof['ticks_per_launch'])
    launches = prof['sampled'] / tpl
    avg_ms = prof['sampled_ms'] / launches
    pos_ticks = prof['sampled_reaches'] / launches          # position-ticks (time-tiled), reach-rows (direct) or reach-ticks (streaming) per launch
    streaming = (72.0 + 16.0 / nsub) * pos_ticks
    if kernel == 'direct':
        alg = pos_ticks * (16.0 + 64.0 / tpl)
        name = f'k_direct (direct row path over column-range tiles, {tpl} rows per task)'
    elif tpl > 1:
        alg = pos_ticks * (16.0 + (TILE_STATE_BYTES + (8.0 if unit else 0.0)) / tpl)
        name = f'k_tile (time-tiled routing over subtree tiles, {tpl} ticks per task)'
    else:
        alg = streaming
        name = 'k_tick (streaming routing)'
    sec = avg_ms * 1e-3
    per_launch = None if not traffic else traffic.get('main_kernel_bytes_per_launch')
    out = {'bound': 'hbm', 'achieved': round(alg / sec / 1e9, 1), 'peak': peak_gbs, 'unit': 'GB/s',
           'frac': round(alg / sec / 1e9 / peak_gbs, 4), 'traffic': None if per_launch is None else round(per_launch),
           'traffic_source': None if not traffic else traffic.get('source'),
           'kernel': name, 'ticks_per_launch': tpl, 'avg_launch_us': round(avg_ms * 1e3, 3),
           'algorithmic_bytes_per_launch': round(alg),
           'algorithmic_bytes_per_position_tick': round(alg / pos_ticks, 3),
           'measured_hbm_gbps': None if per_launch is None else round(per_launch / sec / 1e9, 1),
           'frac_measured_traffic': None if per_launch is None else round(per_launch / sec / 1e9 / peak_gbs, 4),
           'peak_measured_copy': None if copy_gbs is None else round(copy_gbs, 1),
           'frac_of_measured_copy': None if copy_gbs is None else round(alg / sec / 1e9 / copy_gbs, 4),
           'streaming_model_gbps': round(streaming / sec / 1e9, 1),
           'launches_per_pass': prof['launches'], 'launches_sampled': int(launches),
           'pass_region_ms': round(prof['region_ms'], 3)}
    return out



def whole_path(roofline, prof, aux, kern, rate, traffic):
    """Adds the WHOLE path to the `roofline` object (which prices the dominant kernel): every kernel of the timed pass with its
    launches and its average duration between HIP events on the engine's stream (every fourth launch sampled; the direct launches
    all), the bytes per reach-step the path has to move (16: a lateral value in, a discharge out) and -- from the committed
    rocprofv3 counter passes of the same command, while the kernel sources are the ones they were taken with -- the bytes it does
    move, and the two end-to-end fractions of the 8 TB/s peak that follow from the measured rate."""
    if roofline is None:
        return
    main = {'tile': 'k_tile', 'direct': 'k_direct', 'tick': 'k_tick'}[kern]
    kernels = {}
    if prof['brackets'] > 0:
        kernels[main] = {'launches': prof['launches'], 'sampled': prof['brackets'], 'avg_us': round(prof['sampled_ms'] / prof['brackets'] * 1e3, 2)}
    if kern == 'direct':      # prof['launches'] counts the schedule's steps (direct launches, then the skeleton's drain); every direct launch is sampled
        kernels[main]['launches'] = prof['brackets']
    for name, a in aux.items():
        if a['sampled'] > 0:
            kernels[name] = {'launches': a['launches'], 'sampled': a['sampled'], 'avg_us': round(a['sampled_ms'] / a['sampled'] * 1e3, 2)}
    for k in kernels.values():
        k['ms_per_pass'] = round(k['launches'] * k['avg_us'] / 1e3, 2)
    measured = None if traffic is None else traffic.get('bytes_per_reach_step')
    # at the level the driver reads: `frac` prices ONE kernel (roofline.kernel says which) on its algorithmic bytes; these price the
    # whole timed pass -- on the bytes it has to move (16 per reach-step) and on the bytes the counters saw it move
    roofline['frac_end_to_end_compulsory'] = round(16.0 * rate / 1e9 / HBM_PEAK_GBS, 4)
    roofline['bytes_per_reach_step_compulsory'] = 16.0
    roofline['bytes_per_reach_step_measured'] = measured
    roofline['frac_end_to_end_measured'] = None if measured is None else round(measured * rate / 1e9 / HBM_PEAK_GBS, 4)
    # the dominant kernel's measured HBM rate from pass totals: the bytes of ALL its dispatches in the counter passes over the time of
    # all its launches here (round 4 divided the median bytes of a full launch by the average time of all launches)
    total_bytes = None if traffic is None else (traffic.get('pass_bytes') or {}).get(main)
    if total_bytes and main in kernels and kernels[main]['ms_per_pass'] > 0:
        roofline['measured_hbm_gbps'] = round(total_bytes / (kernels[main]['ms_per_pass'] * 1e-3) / 1e9, 1)
        roofline['frac_measured_traffic'] = round(roofline['measured_hbm_gbps'] / HBM_PEAK_GBS, 4)
        roofline['traffic'] = round(total_bytes / max(1, kernels[main]['launches']))      # HBM bytes per launch, averaged over the pass like `achieved`
    roofline['path'] = {
        'kernels': kernels, 'sum_ms_per_pass': round(sum(k['ms_per_pass'] for k in kernels.values()), 2),
        'bytes_per_reach_step_compulsory': 16.0,
        'bytes_per_reach_step_measured': measured,
        'bytes_per_reach_step_by_kernel': None if traffic is None else traffic.get('by_kernel'),
        'traffic_source': None if traffic is None else traffic.get('source'),
        'frac_end_to_end_compulsory': round(16.0 * rate / 1e9 / HBM_PEAK_GBS, 4),
        'frac_end_to_end_measured': None if measured is None else round(measured * rate / 1e9 / HBM_PEAK_GBS, 4)}


def engine_sha16():
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(REPO, 'river_route_amd', 'csrc')
    for name in sorted(f for f in os.listdir(csrc) if f.endswith(('.hip', '.hpp', '.cpp'))):      # every source of librr_hip.so
        with open(os.path.join(csrc, name), 'rb') as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def pmc_traffic(order, n, T, nsub, key=None):
    """HBM bytes of the timed pass from the committed counter passes (profiles/r05_pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE and
    --pmc WRITE_SIZE, separate runs of this bench command, every dispatch of the pass summed per kernel, FETCH_SIZE doubled as the
    micro-architecture guide prescribes for gfx950) -- only for the configuration they were taken on and only while the kernel
    sources are the ones they were taken with."""
    path = PMC_JSON
    order = key or order      # the entry's name: the params order of the headline's network, or 'config2' / 'config4' / 'f32' for the secondary lines
    if nsub != 1 or any(k.startswith(('RR_WAVE', 'RR_TILE', 'RR_DIRECT')) for k in os.environ):
        return None
    try:
        with open(path) as f:
            rec = json.load(f).get(order)
        if not rec or rec.get('reaches') != n or rec.get('runoff_steps') != T:
            return None
        if rec.get('engine_sha16') != engine_sha16():
            return {'source': f'profiles/r05_pmc_traffic.json [{order}] is from other kernel sources ({rec.get("engine_sha16")}): not used'}
        out = {'source': f'profiles/r05_pmc_traffic.json [{order}] (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over the whole year, engine {rec["engine_sha16"]})',
               'bytes_per_reach_step': rec['bytes_per_reach_step'], 'by_kernel': rec['bytes_per_reach_step_by_kernel'],
               'main_kernel_bytes_per_launch': rec.get('main_kernel_bytes_per_full_launch'),
               'pass_bytes': {k: v['hbm_read_bytes'] + v['hbm_write_bytes'] for k, v in rec.get('kernels', {}).items()}}
        return out
    except (OSError, KeyError, ValueError):
        return None


