// LSST_Image launch planner: host code behind ims_plan_* (include/imsim_hip.h).
//
// What the reference does object by object in LSST_ImageBuilder.buildImage (imsim/lsst_image.py:341-368) -- set the stamp up,
// draw it through the sensor, add it to the CCD -- is here ONE prepared list of launches per CCD.  This file derives that list
// from the per-object photon counts and stamp bounds: the numpy planner of imsim_amd/engine.py (Renderer.plan_lsst_image +
// _compile_plan), restated as one native pass so that a CCD of a focal plane costs the host a fraction of a millisecond
// instead of five.  The Python planner stays as the checker (tests/test_device_table.py, tests/test_parity_gpu.py).
//
// Included by imsim_hip.hip inside its extern "C" block's translation unit; uses ims_run_plan, ims_gather_rows and the error
// helpers defined there.
#pragma once

#include <algorithm>
#include <cstring>
#include <numeric>
#include <vector>

namespace ims_planner {

constexpr int ROLE_CHAIN = 0, ROLE_BULK = 1;       // Renderer.STREAMS: chain 0, bulk 1, chain1 2, chain2 3, chain3 4
constexpr int CHAIN_ROLE[4] = { 0, 2, 3, 4 };
constexpr int64_t ALIGN = 256;

struct Launch {                     // one launch table
    int kind = 0;                   // 0 fused render, 1 pool shoot, 2 chain class table
    int64_t n = 0, n_segments = 0, photons = 0;
    int64_t off_index = 0, off_first = 0, off_count = 0, off_bf = 0;      // gather inputs (arena offsets)
    int64_t off_prefix = 0, off_segobj = -1, off_pool = -1;               // segment tables, pool offsets
    int64_t rows_off = 0;                                                 // byte offset of the gathered rows
    int64_t realized_off = -1;                                            // first entry in the realized scratch
    int group = -1;
    ims_render_params_t P;
};

struct Step {                       // one plan item before the addresses are known
    int kind = 0, stream = 0;
    int launch = -1;                // index into launches (render / shoot)
    int first_slot = 0, n_slots = 0;
    int64_t off_tile_prefix = -1, n_tiles = 0;
    int event = 0;
    int chain_begin = 0, n_chains = 0;      // IMS_PLAN_ROUNDS
};

struct ChainDesc {
    int launch = -1;                // the class table
    int first_slot = 0, stream = 0, n_rounds = 0, ev_base = 0;
    std::vector<int32_t> edges;     // slice starts
    std::vector<int64_t> n_phot;    // host, descending
    int64_t off_tile_prefix = -1;
    std::vector<int64_t> tile_prefix_host;
};

struct Group {
    int64_t off_slots = 0;          // arena offset of the slot table (ims_bf_slot_t[n_slots])
    int n_slots = 0;
    int64_t pool_photons = 0;
    std::vector<Step> steps;
    std::vector<ChainDesc> chains;
    // bound form
    std::vector<ims_plan_item_t> items;
    std::vector<ims_chain_t> chain_structs;
    ims_photons_t pool;
};

struct Plan {
    ims_plan_input_t in;
    ims_plan_sizes_t sizes;
    std::vector<uint8_t> arena;
    std::vector<Launch> launches;
    std::vector<Group> groups;          // group 0 may be slot-less (no bright objects: just the fused render)
    std::vector<int64_t> realized_where;    // master row of every realized entry (also in the arena for the device)
    int64_t off_realized_where = -1;
    // bound
    bool bound = false, uploaded = false;
    uint8_t* arena_host = nullptr;
    uint8_t* arena_dev = nullptr;
    uint8_t* rows_dev = nullptr;
    const ims_object_t* master_dev = nullptr;
    double* pool_dev = nullptr;
    double* realized_dev = nullptr;
    std::vector<hipEvent_t> events;
    // a deferred run (ims_plan_run_deferred): everything but the rounds of the top chain is enqueued and not yet joined into
    // the main stream; ims_plans_run_joint runs those rounds together with other plans' and joins
    bool deferred = false;
    unsigned left = 0;                  // bit k: the rounds of chain k have not been run yet
    int unjoined = 0;                   // distinct streams whose end-of-plan events a join has to wait for
    std::vector<hipEvent_t> d_events;   // own events: [0 .. 3] chain k's stream ready, then one per joint run this plan took part in
    size_t d_done_used = 0;             // joint-run events in use since the last join
    ims_sensor_t* d_sensor_dev = nullptr;
    ims_sensor_t* d_sensor_host = nullptr;
    unsigned char* d_changed = nullptr;
    void* d_main = nullptr;
    std::vector<void*> d_streams;
    int32_t n_slots_scalar = 0;         // staging of sensor_dev->n_bf_slots lives in the arena (one per group)
    std::vector<int64_t> off_nslots;

    template <class T> int64_t add(const T* data, int64_t count)
    {
        const int64_t off = ((int64_t)arena.size() + ALIGN - 1) / ALIGN * ALIGN;
        arena.resize((size_t)(off + std::max<int64_t>(count, 1) * (int64_t)sizeof(T)), 0);
        if (count > 0) std::memcpy(arena.data() + off, data, (size_t)count * sizeof(T));
        return off;
    }
    template <class T> int64_t add(const std::vector<T>& v) { return add(v.data(), (int64_t)v.size()); }
};

// a launch table over the objects `sel` (indices into the input arrays) with per-object first photon (relative to the
// object's own phot_first), count and boundary slot
static int make_launch(Plan& pl, int kind, int group, const std::vector<int64_t>& sel, const std::vector<int64_t>& first,
                       const std::vector<int64_t>& count, const std::vector<int32_t>& bf, const std::vector<int64_t>* pool_off,
                       bool realized)
{
    const ims_plan_input_t& in = pl.in;
    Launch L;
    L.kind = kind; L.group = group; L.n = (int64_t)sel.size();
    std::vector<int64_t> rows(sel.size());
    for (size_t k = 0; k < sel.size(); ++k) rows[k] = in.row ? in.row[sel[k]] : sel[k];
    std::vector<int64_t> prefix(sel.size() + 1, 0);
    for (size_t k = 0; k < sel.size(); ++k) {
        prefix[k + 1] = prefix[k] + (count[k] + in.seg_size - 1) / in.seg_size;
        L.photons += count[k];
    }
    L.n_segments = prefix.back();
    L.off_index = pl.add(rows);
    L.off_first = pl.add(first);
    L.off_count = pl.add(count);
    L.off_bf = pl.add(bf);
    L.off_prefix = pl.add(prefix);
    if (L.n_segments > 0) {
        std::vector<int32_t> seg_obj((size_t)L.n_segments);
        for (size_t k = 0; k < sel.size(); ++k)
            std::fill(seg_obj.begin() + prefix[k], seg_obj.begin() + prefix[k + 1], (int32_t)k);
        L.off_segobj = pl.add(seg_obj);
    }
    if (pool_off) L.off_pool = pl.add(*pool_off);
    L.rows_off = pl.sizes.rows_bytes;
    pl.sizes.rows_bytes += std::max<int64_t>(L.n, 1) * (int64_t)sizeof(ims_object_t);
    if (realized && in.want_realized) {
        L.realized_off = (int64_t)pl.realized_where.size();
        pl.realized_where.insert(pl.realized_where.end(), rows.begin(), rows.end());
    }
    std::memset(&L.P, 0, sizeof(L.P));
    pl.launches.push_back(L);
    return (int)pl.launches.size() - 1;
}

static int build(Plan& pl)
{
    const ims_plan_input_t& in = pl.in;
    ims_plan_sizes_t& sz = pl.sizes;
    std::memset(&sz, 0, sizeof(sz));
    if (in.n < 0 || !in.n_phot || (in.n > 0 && !in.stamp)) return set_err(IMS_ERR_ARG, "plan input: n_phot / stamp is NULL");
    if (in.seg_size != 256) return set_err(IMS_ERR_ARG, "plan input: seg_size must be 256");
    if (in.n_class_rounds < 0 || in.n_class_rounds > 3) return set_err(IMS_ERR_ARG, "plan input: at most three class thresholds");
    // objects with photons; the bright ones (own recalculations) apart, brightest first (stable)
    std::vector<int64_t> normal, bright;
    for (int64_t k = 0; k < in.n; ++k) {
        if (in.n_phot[k] <= 0) continue;
        ++sz.n_objects;
        const bool faint = in.faint && in.faint[k];
        if (in.nrecalc > 0 && in.n_phot[k] > in.nrecalc && !faint) bright.push_back(k); else normal.push_back(k);
    }
    std::stable_sort(bright.begin(), bright.end(), [&](int64_t a, int64_t b) { return in.n_phot[a] > in.n_phot[b]; });
    int n_events = in.event_base;
    bool render_done = false;
    auto add_render = [&](Group& g, int gi) {
        std::vector<int64_t> first(normal.size(), 0), count(normal.size());
        std::vector<int32_t> bf(normal.size(), 0);
        for (size_t k = 0; k < normal.size(); ++k) count[k] = in.n_phot[normal[k]];
        Step s;
        s.kind = IMS_PLAN_RENDER; s.stream = ROLE_BULK;
        s.launch = make_launch(pl, 0, gi, normal, first, count, bf, nullptr, true);
        g.steps.push_back(s);
        const Launch& L = pl.launches[s.launch];
        ++sz.n_render_launches; sz.render_photons += L.photons; sz.render_rows += L.n; sz.render_segments += L.n_segments;
        render_done = true;
    };
    size_t start = 0;
    while (start < bright.size()) {
        // the longest run from `start` that fits the scratch cells, the slot table and the photon pool (plan_bf_groups)
        size_t end = start;
        int64_t cells = 0, photons = 0;
        while (end < bright.size()) {
            const int32_t* st = in.stamp + 4 * bright[end];
            const int64_t c = (int64_t)(st[1] - st[0] + 2) * (int64_t)(st[3] - st[2] + 2);
            if (cells + c > in.scratch_cells) break;
            if ((int64_t)(end - start) >= (int64_t)in.slot_capacity - in.n_static_slots) break;
            if (end > start && photons + in.n_phot[bright[end]] > in.max_pool_photons) break;
            cells += c; photons += in.n_phot[bright[end]]; ++end;
        }
        if (end == start) return set_err(IMS_ERR_ARG, "brighter-fatter scratch capacity too small for one stamp; raise SensorSetup.scratch_cells");
        const int gi = (int)pl.groups.size();
        pl.groups.emplace_back();
        Group& g = pl.groups.back();
        const int n = (int)(end - start), n0 = in.n_static_slots;
        std::vector<ims_bf_slot_t> slots((size_t)n);
        std::vector<int64_t> total((size_t)n), offs((size_t)n + 1, 0), n_rounds((size_t)n);
        int64_t cell_off = in.static_cells;
        for (int k = 0; k < n; ++k) {
            const int64_t o = bright[start + k];
            const int32_t* st = in.stamp + 4 * o;
            slots[k].xmin = st[0]; slots[k].ymin = st[2]; slots[k].nx = st[1] - st[0] + 1; slots[k].ny = st[3] - st[2] + 1;
            slots[k].offset = cell_off;
            cell_off += (int64_t)(slots[k].nx + 1) * (slots[k].ny + 1);
            total[k] = in.n_phot[o];
            offs[k + 1] = offs[k] + total[k];
            n_rounds[k] = (total[k] + in.nrecalc - 1) / in.nrecalc;
        }
        g.off_slots = pl.add(slots);
        g.n_slots = n;
        g.pool_photons = offs[n];
        sz.pool_photons = std::max(sz.pool_photons, g.pool_photons);
        {
            const int32_t v = n0 + n;
            pl.off_nslots.push_back(pl.add(&v, 1));
        }
        // chain classes by round count (objects sorted by photon count: classes are ranges)
        std::vector<int> bounds = { 0, n };
        for (int t = 0; t < in.n_class_rounds; ++t) {
            int c = 0;
            while (c < n && n_rounds[c] >= in.class_rounds[t]) ++c;
            if (c > 0 && c < n) bounds.push_back(c);
        }
        std::sort(bounds.begin(), bounds.end());
        bounds.erase(std::unique(bounds.begin(), bounds.end()), bounds.end());
        const int n_classes = (int)bounds.size() - 1;
        if (n_classes > IMS_MAX_CHAINS) return set_err(IMS_ERR_ARG, "too many chain classes");
        // tile prefix of the slots first .. end of the group, one array per class start
        auto tile_prefix = [&](int first, std::vector<int64_t>& host) {
            host.assign((size_t)(n - first) + 1, 0);
            for (int k = first; k < n; ++k)
                host[k - first + 1] = host[k - first] + (int64_t)((slots[k].nx + 1 + 15) / 16) * ((slots[k].ny + 1 + 15) / 16);
            return pl.add(host);
        };
        g.chains.resize((size_t)n_classes);
        for (int c = 0; c < n_classes; ++c) {
            ChainDesc& ch = g.chains[c];
            const int ca = bounds[c], cb = bounds[c + 1];
            ch.off_tile_prefix = tile_prefix(ca, ch.tile_prefix_host);
            Step s;
            s.kind = IMS_PLAN_INIT; s.stream = CHAIN_ROLE[c]; s.first_slot = n0 + ca; s.n_slots = cb - ca;
            s.off_tile_prefix = ch.off_tile_prefix; s.n_tiles = ch.tile_prefix_host[cb - ca];
            g.steps.push_back(s);
        }
        // pool shoots per class in slices of rounds: the first slice on the class's own stream, the others on the bulk stream
        std::vector<Step> bulk_steps;
        for (int c = 0; c < n_classes; ++c) {
            ChainDesc& ch = g.chains[c];
            const int ca = bounds[c], cb = bounds[c + 1];
            const int rounds = (int)n_rounds[ca];
            ch.n_rounds = rounds; ch.first_slot = n0 + ca; ch.stream = CHAIN_ROLE[c];
            // slices of rounds: fine at the start, so that a chain that runs beside its own shoots starts at once and never waits
            // long; with coarse_slices (a focal plane's deferred plans: the rounds start only when the whole batch is enqueued)
            // round 0 and then everything else -- two launches per class instead of up to six, each wide enough to fill the device
            std::vector<int32_t> edges = { 0 };
            if (in.coarse_slices) { if (1 < rounds) edges.push_back(1); }
            else for (int e : { 1, 3, 8, 20, 60 }) if (e < rounds) edges.push_back(e);
            edges.push_back(rounds);
            ch.ev_base = n_events;
            n_events += (int)edges.size() - 1;
            for (size_t k = 0; k + 1 < edges.size(); ++k) {
                const int64_t ra = edges[k], rb = edges[k + 1];
                std::vector<int64_t> sel, first, count, pool_off;
                std::vector<int32_t> bf;
                for (int j = ca; j < cb; ++j) {
                    const int64_t lo = std::min<int64_t>(total[j], ra * in.nrecalc), hi = std::min<int64_t>(total[j], rb * in.nrecalc);
                    if (hi <= lo) continue;
                    sel.push_back(bright[start + j]); first.push_back(lo); count.push_back(hi - lo);
                    bf.push_back(n0 + j); pool_off.push_back(offs[j] + lo);
                }
                Step s;
                s.kind = IMS_PLAN_SHOOT_POOL; s.stream = (k == 0) ? ch.stream : ROLE_BULK;
                s.launch = make_launch(pl, 1, gi, sel, first, count, bf, &pool_off, false);
                const Launch& L = pl.launches[s.launch];
                ++sz.n_shoot_launches; sz.shoot_photons += L.photons; sz.shoot_rows += L.n; sz.shoot_segments += L.n_segments;
                if (k == 0) g.steps.push_back(s);
                else {
                    // (A "head start" of the top class -- the bulk stream holding back the class's next pool slice until the chain has
                    // consumed the one before -- was built in round 4, but its waits were enqueued ahead of the records they were
                    // meant to wait for, so it never throttled anything and its "no change" measured nothing; removed in round 5.)
                    bulk_steps.push_back(s);
                    Step r;
                    r.kind = IMS_PLAN_RECORD; r.stream = ROLE_BULK; r.event = ch.ev_base + (int)k;
                    bulk_steps.push_back(r);
                }
            }
            edges.pop_back();
            if ((int)edges.size() > IMS_MAX_CHAIN_EDGES) return set_err(IMS_ERR_ARG, "too many pool slices for one chain");
            ch.edges = edges;
        }
        g.steps.insert(g.steps.end(), bulk_steps.begin(), bulk_steps.end());
        // the ordinary objects' fused launch right behind the pool slices on the bulk stream, ahead of the rounds
        if (!normal.empty() && !render_done) add_render(g, gi);
        // the rounds of all classes as ONE item
        for (int c = 0; c < n_classes; ++c) {
            ChainDesc& ch = g.chains[c];
            const int ca = bounds[c], cb = bounds[c + 1];
            std::vector<int64_t> sel, first(cb - ca, 0), count, start_off;
            std::vector<int32_t> bf;
            for (int j = ca; j < cb; ++j) {
                sel.push_back(bright[start + j]); count.push_back(total[j]); bf.push_back(n0 + j); start_off.push_back(offs[j]);
                ch.n_phot.push_back(total[j]);
            }
            ch.launch = make_launch(pl, 2, gi, sel, first, count, bf, &start_off, true);
            sz.chain_rows += cb - ca;
            sz.n_round_launches += ch.n_rounds;                 // one pixel-search launch per round of the class
        }
        Step s;
        s.kind = IMS_PLAN_ROUNDS; s.stream = 0; s.chain_begin = 0; s.n_chains = n_classes;
        g.steps.push_back(s);
        start = end;
    }
    if (!normal.empty() && !render_done) {
        pl.groups.emplace_back();
        add_render(pl.groups.back(), (int)pl.groups.size() - 1);
    }
    if (n_events - in.event_base > 1000) return set_err(IMS_ERR_ARG, "a plan may use at most 1000 library events");
    if (!pl.realized_where.empty()) pl.off_realized_where = pl.add(pl.realized_where);
    sz.realized_count = (int64_t)pl.realized_where.size();
    sz.arena_bytes = ((int64_t)pl.arena.size() + ALIGN - 1) / ALIGN * ALIGN;
    pl.arena.resize((size_t)sz.arena_bytes, 0);
    sz.rows_bytes = std::max<int64_t>(sz.rows_bytes, (int64_t)sizeof(ims_object_t));
    sz.n_groups = (int)pl.groups.size();
    sz.n_events = n_events - in.event_base;
    return IMS_OK;
}

}  // namespace ims_planner
