// render_tiles.hpp — launch code of one precision: rttnw_render_tiles_device's body (passes, kernel selection, workspace),
// the per-bounce probe and the un-tile launch, as templates over the arithmetic type.
#pragma once
#include "trace_kernels.hpp"

#ifndef RT_TINY_TREE_STEPS
#define RT_TINY_TREE_STEPS 3 // node steps per walk trip of the lane-owns-path kernel for top trees of <= 16 nodes (RT_NODE_STEPS otherwise)
#endif
namespace rt {
inline namespace RT_ARITH_NS {

// prepare_only: upload the scene on first use and grow every workspace buffer this render will need (blocking hipMalloc /
// hipMemcpy / hipFree calls), launch nothing — rttnw_render_multi does that for ALL its ranks before the first launch, so
// that no allocation (a device-wide synchronisation) sits between two ranks' kernels.
template <typename R>
int render_tiles_t(::rttnw_scene* s, DeviceState* d, const rttnw_camera_desc* cam, const rttnw_params* p, void* d_packed, hipStream_t stream,
                   rttnw_stats* stats, bool sync_for_stats, bool prepare_only) {
    HIP_TRY(hipSetDevice(d->device));
    const FlatScene* flat_p = &s->flat;
    DeviceScene<R>* ds_p = &scene_of<R>(d);
#if defined(RT_STRICT_F64)
    // the IEEE-strict build walks the lowering that tests every object in the reference's frame (render_api.cpp reference_frame_scene)
    if (int rc = reference_frame_scene(s, flat_p)) return rc;
    if (flat_p != &s->flat) ds_p = &d->s64_ref;
#endif
    const FlatScene& flat = *flat_p;
    DeviceScene<R>& ds = *ds_p;
    if (!ds.ready)
        if (int rc = ds.upload(flat)) return rc;

    rttnw_tile_layout L;
    fill_layout(p->width, p->height, p->tile_world, L);
    RenderConsts rc{};
    rc.width = p->width; rc.height = p->height; rc.spp = p->spp; rc.max_depth = p->max_depth;
    rc.tiles_x = L.tiles_x; rc.tiles_y = L.tiles_y; rc.n_tiles = L.n_tiles;
    rc.tile_rank = p->tile_rank; rc.tile_world = p->tile_world;
    rc.my_tiles = L.n_tiles > p->tile_rank ? (L.n_tiles - p->tile_rank + p->tile_world - 1) / p->tile_world : 0;
    rc.quirks = p->quirks; rc.seed = p->seed; rc.stack_depth = flat.stack_depth;
    rc.profile = p->collect_counters;
    rc.sample_begin = p->sample_begin;
    rc.scene_flags = flat.moving.empty() ? SCENE_NO_TIME : 0u;
    rc.inv_width = 1.0 / double(p->width); rc.inv_height = 1.0 / double(p->height);
    rc.div_tiles_x = make_fastdiv(std::max<uint32_t>(1u, rc.tiles_x));
    // The render's chunk schedule (a function of spp alone) and how many of its chunks one launch traces (rt_types.hpp
    // launch_chunks: the chunk sums of a launch stay within the device's budget; the resolve step continues every pixel's
    // chain, so the image does not depend on the split).
    plan_chunks(rc, p->spp, p->spp_chunk);
    const uint32_t total_chunks = rc.n_chunks;
    // (if the device cannot give the workspace — other tenants of its memory — the budget is halved, down to 1 GiB: more launches, same image)
    // The budget that worked is REMEMBERED (d->chunk_budget): a later render does not retry the allocation that failed, and the
    // buffer in hand is released only once a larger one has been obtained — or, if none can be, simply used: the launch split
    // then follows ITS size.
    uint32_t per_launch = 0;
    for (uint64_t budget = d->chunk_budget;; budget /= 2) {
        per_launch = launch_chunks(uint64_t(rc.my_tiles) * 64, 3 * sizeof(R), total_chunks, budget);
        const size_t want = std::max<size_t>(size_t(rc.my_tiles) * 64 * std::min(per_launch, total_chunks), 1) * 3 * sizeof(R);
        if (d->partial_bytes >= want && d->partial) break;
        void* bigger = nullptr;
        const hipError_t e = hipMalloc(&bigger, std::max<size_t>(want, 16));
        if (e == hipSuccess) {
            if (d->partial) (void)hipFree(d->partial);
            d->partial = bigger;
            d->partial_bytes = want;
            if (!getenv("RTTNW_CHUNK_SUM_BUDGET")) d->chunk_budget = budget;
            break;
        }
        (void)hipGetLastError(); // clear the sticky out-of-memory
        const size_t one_group = size_t(rc.my_tiles) * 64 * std::min<uint32_t>(16u, total_chunks) * 3 * sizeof(R);
        // (not under RTTNW_CHUNK_SUM_BUDGET: launch_chunks() would size the launches by the environment's budget again, not by the buffer in hand)
        if (d->partial && d->partial_bytes >= one_group && !getenv("RTTNW_CHUNK_SUM_BUDGET")) { // no larger buffer to be had: split the render by the one in hand
            per_launch = launch_chunks(uint64_t(rc.my_tiles) * 64, 3 * sizeof(R), total_chunks, d->partial_bytes);
            if (!getenv("RTTNW_CHUNK_SUM_BUDGET")) d->chunk_budget = std::max<uint64_t>(d->partial_bytes, 1ull << 30);
            break;
        }
        if (budget <= (1ull << 30) || getenv("RTTNW_CHUNK_SUM_BUDGET")) { set_last_error(std::string("render: no memory for the chunk sums: ") + hipGetErrorString(e)); return RTTNW_ERR_HIP; }
    }

    CameraRec<double> cam64;
    make_camera(cam->lookfrom, cam->lookat, cam->view_up, cam->vertical_fov, cam->aspect_ratio, cam->aperture,
                cam->focus_distance, cam->open_time, cam->close_time, cam64);
    const CameraRec<R> camr = narrow_camera<R>(cam64);

    const bool count = p->collect_counters != 0;
    // Two forms of the same loop (DESIGN.md "Kernels"): measured on MI355X the lane-owns-a-path form wins on shallow
    // scenes (cornell_box, final_scene: <= ~1k nodes), the decoupled form on deep BVHs where traversal lengths vary
    // most (1M spheres).  RTTNW_KERNEL=plain|plainglobal|wave overrides the choice (experiments only).
    const char* kv = getenv("RTTNW_KERNEL");
    // (crossover measured on spheres_1m-like scenes of 4e3 - 1e5 spheres, 512x512 spp 256, lane-owns-path against decoupled, Msamples/s.  Round 3: f32 at
    // ~24 k 4-wide nodes, f64 at ~50 k.  After round 4 — quantised records, no instance code, 13-real path slots, the f64 unit split — the decoupled kernel
    // takes over much earlier: f32 5.7 k nodes 7158 / 6646, 9.9 k 5272 / 5403, 15.4 k 3917 / 4460, 28.3 k 2273 / 3184; f64 9.9 k 4776 / 4039, 15.4 k
    // 3559 / 3340, 19.6 k 2818 / 2898, 28.3 k 1979 / 2332, 50.9 k 1223 / 1722)
    // (round 5 — asynchronous shade phases in the lane-owns-path kernel, the f64 decoupled kernel at three blocks per CU: f32 9.9 k nodes 5441 / 5350,
    // 15.4 k 4180 / 4483, 19.6 k 3400 / 3911; f64 9.9 k 5168 / 4898, 15.4 k 3960 / 4153, 19.6 k 3166 / 3667; RTTNW_F64_STRICT 15.4 k 4015 / 4110 —
    // profiles/r05/README.md: both cross at ~13 k records)
    // (later in round 5 — the decoupled kernel keeps a slot's ray in LDS, +5 .. 9 %: f32 5.7 k nodes 7216 / 7620, 7.6 k 6101 / 6840, 9.9 k 5410 / 6371; f64
    // 5.7 k 6974 / 6577, 7.6 k 5919 / 5817, 9.9 k 5150 / 5318, 15.4 k 3941 / 4450; RTTNW_F64_STRICT 7.6 k 5945 / 5783, 9.9 k 5195 / 5224: f32 crosses
    // at ~5 k records, f64 at ~9 k)
    bool plain = flat.total_nodes4() < (sizeof(R) == 4 ? 5000u : 9000u);
    if (kv && (std::strcmp(kv, "plain") == 0 || std::strcmp(kv, "plainglobal") == 0)) plain = true;
    if (kv && std::strcmp(kv, "wave") == 0) plain = false;
    if (!prepare_only) HIP_TRY(hipMemsetAsync(d->job_counter, 0, sizeof(unsigned long long) + sizeof(DeviceCounters), stream));
    DeviceCounters* dc = reinterpret_cast<DeviceCounters*>(d->job_counter + 1);
    auto persistent_grid = [&](const void* kernel, size_t lds_bytes, size_t waves_needed, size_t& grid, int block = TRACE_BLOCK) -> int {
        if (lds_bytes > 160 * 1024) { set_last_error("render: queues + traversal stacks do not fit in LDS"); return RTTNW_ERR_UNSUPPORTED; }
        HIP_TRY(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, int(lds_bytes)));
        int blocks_per_cu = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks_per_cu, kernel, block, lds_bytes));
        // (the runtime's answer counts LDS in finer units than the hardware allocates it in: trace_kernels.hpp lds_blocks_per_cu)
        blocks_per_cu = std::max(1, std::min(std::min(blocks_per_cu, 8), int(lds_blocks_per_cu(uint32_t(lds_bytes)))));
        // what the chip holds at once (no inter-workgroup dependency, so a little over-subscription is harmless),
        // but never more waves than there is work for
        const size_t waves_per_block = size_t(block) / 64;
        grid = std::max<size_t>(1, std::min<size_t>(size_t(d->num_cus) * blocks_per_cu, (waves_needed + waves_per_block - 1) / waves_per_block));
        return 0;
    };
    // stack entries beyond the LDS-resident ones, for every thread of a launch
    auto grow_spill = [&](size_t threads, uint32_t lds_entries = LDS_STACK_ENTRIES) -> int {
        const size_t extra = rc.stack_depth > lds_entries ? rc.stack_depth - lds_entries : 0;
        return grow(&d->spill, &d->spill_bytes, std::max<size_t>(threads * extra, 1) * sizeof(int32_t));
    };
    bool three_steps = false; // the lane-owns-path kernel's instantiation with three node steps per trip (tiny top trees)
    bool no_inst = false;     // the decoupled kernel's instantiation for scenes without instance records
    bool no_time = false;     // ... and the LEAN flavour of the frame-less instantiations (no moving sphere, no medium, solid colours only: rt_core.hpp SHAPES_*_NT)
    // One pass: trace kernel over the pass's jobs, then the resolve step.
    auto trace_pass = [&]() -> int {
        const size_t n_jobs = rc.n_jobs;
        if (plain) {
            // Small scenes: node array in LDS, in ONE large block per CU so that nodes + all the lanes' stacks fit in 160 KB:
            // 1024 threads in both precisions (4 waves/SIMD at <= 128 VGPRs; RT_F64_BLOCK: the f64 code spills ~26 registers to get there)
            constexpr int LDS_BLOCK = sizeof(R) == 4 ? 1024 : RT_F64_BLOCK;
            const uint32_t n4 = flat.total_nodes4();
            const bool want_lds = !(kv && std::strcmp(kv, "plainglobal") == 0) && lds_form_bytes(n4, rc.stack_depth, LDS_BLOCK) <= 160 * 1024;
            rc.lds_nodes = want_lds ? n4 : 0u;
            // the scene's Perlin tables ride along in LDS when they fit behind the stacks (trace_kernel_plain)
            const size_t n_perlin = flat.perlin_vec.size() / 768u;
            size_t perlin_bytes = lds_perlin_bytes(n_perlin, sizeof(R));
            if (want_lds && n_perlin > 0 && n_perlin < 256 && n4 <= LDS_NODES_MASK && lds_form_bytes(n4, rc.stack_depth, LDS_BLOCK) + perlin_bytes <= 160 * 1024)
                rc.lds_nodes |= uint32_t(n_perlin) << LDS_PERLIN_SHIFT;
            else
                perlin_bytes = 0;
            // ... and so do the record arrays of the leaf steps, each if it still fits (chains, rectangles, moving spheres, cubes)
            for (int k = 0; k < 6; ++k) rc.lds_recs[k] = 0;
            if (want_lds) {
                perlin_bytes = lds_pad32(perlin_bytes);
                const size_t counts[4] = {flat.insts.size(), flat.rects.size(), flat.moving.size(), flat.boxes.size()};
                const size_t sizes[4] = {sizeof(InstanceRec<R>), sizeof(RectRec<R>), sizeof(MovingSphereRec<R>), sizeof(BoxRec<R>)};
                for (int k = 0; k < 4; ++k) {
                    const size_t bytes = lds_pad32(counts[k] * sizes[k]);
                    if (counts[k] == 0 || lds_form_bytes(n4, rc.stack_depth, LDS_BLOCK) + perlin_bytes + bytes > 160 * 1024) continue;
                    rc.lds_recs[k] = uint32_t(counts[k]);
                    perlin_bytes += bytes;
                }
                {
                    const size_t bytes = lds_pad32(flat.sphere_mat.size() * 4);
                    if (!flat.sphere_mat.empty() && !flat.sphere_mat_is_index && lds_form_bytes(n4, rc.stack_depth, LDS_BLOCK) + perlin_bytes + bytes <= 160 * 1024) {
                        rc.lds_recs[4] = uint32_t(flat.sphere_mat.size());
                        perlin_bytes += bytes;
                    }
                }
            }
            const int block = want_lds ? LDS_BLOCK : TRACE_BLOCK;
            const bool gen = flat.needs_general; // rare graph shapes: the instantiation that carries their code
            // a top tree of one or two levels (cornell_box: 6 nodes; its walks are mostly entered instances) takes three node steps per trip
            // ... and a scene whose walk never changes frames the instantiation without instance code (rt_core.hpp SHAPES_NONE: final_scene — its one
            // instance record is the bare chain of the cluster's world-space copies)
            const bool lds_no_inst = want_lds && !gen && !count && !flat.walk_changes_frames;
            three_steps = want_lds && n4 <= 16u && RT_NODE_STEPS == 2;
            const bool tiny_tree = three_steps && !count; // (the counting variant's tallied loop is written for two: same steps per lane, same counters)
            // (... and, of those two, the LEAN flavour where the scene holds no moving sphere, no medium and only solid colours: rt_core.hpp SHAPES_*_NT)
            no_time = flat.lean();
            const void* kernel =
                lds_no_inst && !flat.has_instance_leaves && no_time ? (tiny_tree ? (const void*)trace_kernel_plain<R, false, LDS_BLOCK, true, SHAPES_NONE_NT, RT_TINY_TREE_STEPS> : (const void*)trace_kernel_plain<R, false, LDS_BLOCK, true, SHAPES_NONE_NT>) :
                lds_no_inst && !flat.has_instance_leaves ? (tiny_tree ? (const void*)trace_kernel_plain<R, false, LDS_BLOCK, true, SHAPES_NONE, RT_TINY_TREE_STEPS> : (const void*)trace_kernel_plain<R, false, LDS_BLOCK, true, SHAPES_NONE>) :
                lds_no_inst && no_time ? (tiny_tree ? (const void*)trace_kernel_plain<R, false, LDS_BLOCK, true, SHAPES_SINGLE_NT, RT_TINY_TREE_STEPS> : (const void*)trace_kernel_plain<R, false, LDS_BLOCK, true, SHAPES_SINGLE_NT>) :
                lds_no_inst ? (tiny_tree ? (const void*)trace_kernel_plain<R, false, LDS_BLOCK, true, SHAPES_SINGLE, RT_TINY_TREE_STEPS> : (const void*)trace_kernel_plain<R, false, LDS_BLOCK, true, SHAPES_SINGLE>) :
                tiny_tree ? (gen ? (const void*)trace_kernel_plain<R, false, LDS_BLOCK, true, SHAPES_GENERAL, RT_TINY_TREE_STEPS> : (const void*)trace_kernel_plain<R, false, LDS_BLOCK, true, SHAPES_FAST, RT_TINY_TREE_STEPS>) :
                want_lds ? (count ? (gen ? (const void*)trace_kernel_plain<R, true, LDS_BLOCK, true, true> : (const void*)trace_kernel_plain<R, true, LDS_BLOCK, true, false>)
                                  : (gen ? (const void*)trace_kernel_plain<R, false, LDS_BLOCK, true, true> : (const void*)trace_kernel_plain<R, false, LDS_BLOCK, true, false>))
                         : (count ? (gen ? (const void*)trace_kernel_plain<R, true, TRACE_BLOCK, false, true> : (const void*)trace_kernel_plain<R, true, TRACE_BLOCK, false, false>)
                                  : (gen ? (const void*)trace_kernel_plain<R, false, TRACE_BLOCK, false, true> : (const void*)trace_kernel_plain<R, false, TRACE_BLOCK, false, false>));
            const size_t lds_bytes = lds_form_bytes(want_lds ? n4 : 0u, rc.stack_depth, uint32_t(block)) + perlin_bytes;
            if (lds_bytes > 160 * 1024) { set_last_error("render: traversal stacks do not fit in LDS"); return RTTNW_ERR_UNSUPPORTED; }
            HIP_TRY(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, int(lds_bytes)));
            int blocks_per_cu = 0;
            HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks_per_cu, kernel, block, lds_bytes));
            blocks_per_cu = std::max(1, std::min(blocks_per_cu, 8));
            const size_t waves_per_block = size_t(block) / 64;
            const size_t grid = std::max<size_t>(1, std::min<size_t>(size_t(d->num_cus) * blocks_per_cu, ((n_jobs + 63) / 64 + waves_per_block - 1) / waves_per_block));
            if (int g = grow_spill(grid * size_t(block))) return g;
            if (n_jobs > 0 && !prepare_only) {
                R bg0 = R(p->background[0]), bg1 = R(p->background[1]), bg2 = R(p->background[2]), tmin = R(p->t_min);
                R* part = (R*)d->partial;
                unsigned long long* jc = d->job_counter;
                SceneView<R> view = ds.view;
                CameraRec<R> camv = camr;
                int32_t* sp = (int32_t*)d->spill;
                void* args[] = {&view, &camv, &rc, &bg0, &bg1, &bg2, &tmin, &part, &jc, &dc, &sp};
                HIP_TRY(hipLaunchKernel(kernel, dim3(uint32_t(grid)), dim3(block), args, lds_bytes, stream));
            }
        } else {
            const bool gen = flat.needs_general;
            if constexpr (wave_walks_quantised<R>()) {
                if (int q4 = ds.ensure_quant4(flat)) return q4; // this kernel walks the quantised records: made here, on the device, once
            }
            // (a scene without any instance record takes the instantiation whose walk never changes frames, rt_core.hpp SHAPES_NONE)
            no_inst = !gen && !count && !flat.has_instance_leaves;
            no_time = flat.lean();
            auto kernel = count ? (gen ? trace_kernel<R, true, SHAPES_GENERAL> : trace_kernel<R, true, SHAPES_FAST>)
                                : (gen ? trace_kernel<R, false, SHAPES_GENERAL>
                                       : (no_inst ? (no_time ? trace_kernel<R, false, SHAPES_NONE_NT> : trace_kernel<R, false, SHAPES_NONE>) : trace_kernel<R, false, SHAPES_FAST>));
            // (the LEAN flavour: ONE block per CU of as many waves as its LDS holds — 13 in f64, where three 4-wave blocks make 12; RTTNW_WAVE_BLOCK=<threads>: experiments)
            const bool lean_kernel = !count && !gen && no_inst && no_time;
            const uint32_t wave_bytes = wave_lds_bytes<R>(rc.stack_depth, lean_kernel);
            int wblock = lean_kernel ? int(wave_block_waves(wave_bytes)) * 64 : TRACE_BLOCK;
            // (clamped to what a CU's LDS holds: 16 f64 waves would ask for 198 KB and fail the whole render instead of running with 13)
            if (const char* e = getenv("RTTNW_WAVE_BLOCK")) { const int v = atoi(e); if (v >= 64 && v % 64 == 0 && v <= (lean_kernel ? 1024 : TRACE_BLOCK)) wblock = lean_kernel ? std::min(v, wblock) : v; }
            const size_t lds_bytes = size_t(wave_bytes) * size_t(wblock / 64);
            size_t grid = 1;
            if (int g = persistent_grid((const void*)kernel, lds_bytes, (n_jobs + SLOTS_PER_WAVE - 1) / SLOTS_PER_WAVE, grid, wblock)) return g;
            const size_t n_slots = grid * size_t(wblock / 64) * SLOTS_PER_WAVE;
            if (int g = grow(&d->pool_r, &d->pool_r_bytes, n_slots * PR_COUNT * sizeof(R))) return g;
            if (int g = grow(&d->pool_u, &d->pool_u_bytes, n_slots * PU_COUNT * sizeof(uint32_t))) return g;
            if (int g = grow_spill(grid * size_t(wblock), wave_stack_entries<R>())) return g;
            if (n_jobs > 0 && !prepare_only) {
                hipLaunchKernelGGL(kernel, dim3(uint32_t(grid)), dim3(uint32_t(wblock)), lds_bytes, stream, ds.decoupled_view(), camr, rc, R(p->background[0]),
                                   R(p->background[1]), R(p->background[2]), R(p->t_min), (R*)d->partial, d->job_counter, dc, (R*)d->pool_r,
                                   (uint32_t*)d->pool_u, uint32_t(n_slots), (int32_t*)d->spill);
                HIP_TRY(hipGetLastError());
            }
        }
        return 0;
    };
    if (stats && !prepare_only) HIP_TRY(hipEventRecord(d->ev0, stream));
    for (uint32_t c0 = 0; c0 < total_chunks; c0 += per_launch) {
        rc.chunk_base = c0;
        rc.n_chunks = std::min(per_launch, total_chunks - c0);
        const bool first = c0 == 0, last = c0 + per_launch >= total_chunks;
        if (!plan_jobs(rc)) { set_last_error("render: more than 2^32 jobs in a launch"); return RTTNW_ERR_UNSUPPORTED; }
        if (int g = grow(&d->partial, &d->partial_bytes, std::max<size_t>(size_t(rc.jobs_per_chunk) * rc.n_chunks, 1) * 3 * sizeof(R))) return g;
        if (!first && !prepare_only) HIP_TRY(hipMemsetAsync(d->job_counter, 0, sizeof(unsigned long long), stream)); // the job counter only: statistics add up
        if (int g = trace_pass()) return g;
        if (prepare_only) continue;
        if (stats && last) HIP_TRY(hipEventRecord(d->ev1, stream));
        hipLaunchKernelGGL(resolve_kernel<R>, dim3((L.pixels_per_rank + 255) / 256), dim3(256), 0, stream, (const R*)d->partial,
                           (R*)d_packed, rc, L.pixels_per_rank, uint32_t(first), uint32_t(last), p->spp);
        HIP_TRY(hipGetLastError());
    }
    if (prepare_only) return RTTNW_OK;

    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        if (sync_for_stats) {
            HIP_TRY(hipStreamSynchronize(stream));
            float ms = 0;
            HIP_TRY(hipEventElapsedTime(&ms, d->ev0, d->ev1));
            stats->kernel_ms = ms;
        }
        // samples traced by this rank: pixels of its tiles that lie inside the image
        uint64_t px_count = 0;
        for (uint32_t t = 0; t < rc.my_tiles; ++t) {
            uint32_t tx, ty;
            tile_unpermute(rc.tile_rank + t * rc.tile_world, rc.tiles_x, tx, ty);
            uint32_t w = std::min(8u, rc.width - tx * 8), h = std::min(8u, rc.height - ty * 8);
            px_count += uint64_t(w) * h;
        }
        stats->samples = px_count * rc.spp;
        if (count && sync_for_stats) {
            DeviceCounters hc;
            HIP_TRY(hipMemcpy(&hc, dc, sizeof(hc), hipMemcpyDeviceToHost));
            stats->rays = hc.rays; stats->nodes_visited = hc.nodes; stats->prims_tested = hc.prims; stats->texel_fetches = hc.texels;
            if (getenv("RTTNW_DEBUG_SCHED")) debug_print_sched(hc, plain, rc.profile, stats->samples); // (debug_sched.cpp)
        }
        stats->n_nodes = flat.total_nodes4();
        stats->n_prims = flat.n_prims_in_bvh;
        stats->scene_bytes = uint32_t(std::min<size_t>(ds.bytes, 0xFFFFFFFFu));
        // which kernel form ran: bit 0 = decoupled (else lane-owns-path), bit 1 = node records resident in LDS (the form bench.py's
        // roofline calls issue-bound)
        stats->reserved = (plain ? 0u : 1u) | (plain && rc.lds_nodes != 0u ? 2u : 0u) | (plain && three_steps ? 4u : 0u) | (!flat.needs_general && (plain ? !flat.walk_changes_frames && rc.lds_nodes != 0u : !flat.has_instance_leaves) ? 8u : 0u) |
                          (!flat.needs_general && plain && !flat.walk_changes_frames && rc.lds_nodes != 0u && flat.has_instance_leaves ? 16u : 0u);
        if ((stats->reserved & 8u) != 0u && flat.lean()) stats->reserved |= 32u; // bit 5: ... in the LEAN flavour (rt_core.hpp SHAPES_*_NT)
        if (!plain && ds.interleaved) stats->reserved |= 64u;                      // bit 6: the decoupled kernel walked the interleaved node + sphere buffer (render_common.hpp)
    }
    return RTTNW_OK;
}

template <typename R>
int probe_path_t(::rttnw_scene* s, const rttnw_camera_desc* cam, const rttnw_params* p, uint32_t px, uint32_t row, uint32_t sample,
                 double* out, uint32_t max_out) {
    DeviceState* d = s->device;
    HIP_TRY(hipSetDevice(d->device));
    const FlatScene* flat_p = &s->flat;
    DeviceScene<R>* ds_p = &scene_of<R>(d);
#if defined(RT_STRICT_F64)
    if (int rc = reference_frame_scene(s, flat_p)) return rc;
    if (flat_p != &s->flat) ds_p = &d->s64_ref;
#endif
    const FlatScene& flat = *flat_p;
    DeviceScene<R>& ds = *ds_p;
    if (!ds.ready)
        if (int rc = ds.upload(flat)) return rc;
    RenderConsts rc{};
    rc.width = p->width; rc.height = p->height; rc.spp = p->spp; rc.max_depth = p->max_depth;
    rc.quirks = p->quirks; rc.seed = p->seed; rc.stack_depth = flat.stack_depth;
    rc.inv_width = 1.0 / double(p->width); rc.inv_height = 1.0 / double(p->height);
    CameraRec<double> cam64;
    make_camera(cam->lookfrom, cam->lookat, cam->view_up, cam->vertical_fov, cam->aspect_ratio, cam->aperture,
                cam->focus_distance, cam->open_time, cam->close_time, cam64);
    DevBuf<double> d_out; // released on every exit path
    DevBuf<int32_t> d_n;
    if (int r = d_out.upload(std::vector<double>(size_t(max_out) * PROBE_STRIDE + 4, 0.0))) return r;
    if (int r = d_n.upload(std::vector<int32_t>(1, 0))) { d_out.release(); return r; }
    struct Release { DevBuf<double>& a; DevBuf<int32_t>& b; ~Release() { a.release(); b.release(); } } release{d_out, d_n};
    DevBuf<int32_t> d_spill;
    if (int r = d_spill.upload(std::vector<int32_t>(std::max<size_t>(rc.stack_depth, 1), 0))) { d_out.release(); d_n.release(); return r; }
    struct Release2 { DevBuf<int32_t>& a; ~Release2() { a.release(); } } release2{d_spill};
    const size_t lds = size_t(LDS_STACK_ENTRIES + 1) * 64 * sizeof(int32_t);
    hipLaunchKernelGGL(probe_path_kernel<R>, dim3(1), dim3(64), lds, 0, ds.view, narrow_camera<R>(cam64), rc, R(p->t_min), px, row,
                       sample, d_out.p, max_out, d_n.p, d_spill.p);
    HIP_TRY(hipGetLastError());
    int32_t n = 0;
    HIP_TRY(hipMemcpy(&n, d_n.p, sizeof(n), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out, d_out.p, size_t(n) * PROBE_STRIDE * sizeof(double), hipMemcpyDeviceToHost));
    // radiance of the sample (path_step loop) is returned after the last possible bounce record
    HIP_TRY(hipMemcpy(out + size_t(max_out) * PROBE_STRIDE, d_out.p + size_t(max_out) * PROBE_STRIDE, 4 * sizeof(double), hipMemcpyDeviceToHost));
    return n;
}

template <typename R>
int untile_launch(uint32_t width, uint32_t height, uint32_t world, const void* d_gathered, void* d_linear_rgb, uint8_t* d_rgba8, hipStream_t stream) {
    rttnw_tile_layout L;
    fill_layout(width, height, world, L);
    dim3 block(32, 8), grid((width + 31) / 32, (height + 7) / 8);
    hipLaunchKernelGGL(untile_kernel<R>, grid, block, 0, stream, (const R*)d_gathered, (R*)d_linear_rgb, d_rgba8, width, height, L.tiles_x, world,
                       L.pixels_per_rank);
    HIP_TRY(hipGetLastError());
    return RTTNW_OK;
}

// what a precision's translation unit instantiates
#define RT_INSTANTIATE_PRECISION(R)                                                                                                             \
    template int render_tiles_t<R>(::rttnw_scene*, DeviceState*, const rttnw_camera_desc*, const rttnw_params*, void*, hipStream_t, rttnw_stats*, \
                                   bool, bool);                                                                                                 \
    template int probe_path_t<R>(::rttnw_scene*, const rttnw_camera_desc*, const rttnw_params*, uint32_t, uint32_t, uint32_t, double*, uint32_t); \
    template int untile_launch<R>(uint32_t, uint32_t, uint32_t, const void*, void*, uint8_t*, hipStream_t);

} // namespace RT_ARITH_NS
} // namespace rt
