/*
 * fyprt.h — C ABI of the MI355X-native trace + shade path (libfyprt.so).
 *
 * This is the drop-in boundary for ONE path of Savasstion/FYPRayTracer: everything
 * `Renderer::Render` does between "scene/camera are on the host" and "RGBA8 + float4
 * accumulation are back on the host" (reference: FYPRayTracer/src/Classes/Core/Renderer.cu:13-284
 * and the 11 __global__ kernels it launches, Renderer.cu:2431-2900).  Plain pointers and sizes
 * only; no C++, HIP or torch types cross this boundary.  Each entry point cites the
 * reference interface it replaces.  The reference-side binding a maintainer would add is
 * shown in INTEGRATION.md.
 *
 * Error convention: every call returns FYPRT_OK (0) or a negative FYPRT_E* code and
 * records a message retrievable with fyprt_last_error().  (The reference prints
 * cudaGetErrorString to stderr and keeps going, Renderer.cu:29-47; the C++ facade in
 * fypraytracer_amd/host/ reproduces that on top of these codes.)
 *
 * Threading: one context per host thread and per GPU; a context owns one HIP stream.
 */
#ifndef FYPRT_H
#define FYPRT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FYPRT_OK 0
#define FYPRT_EINVAL (-1)   /* bad argument / inconsistent scene description            */
#define FYPRT_EHIP (-2)     /* a HIP runtime call failed (message has hipGetErrorString) */
#define FYPRT_ESTATE (-3)   /* call order violated (render before resize/scene/camera)   */
#define FYPRT_ENOLIGHT (-4) /* light-based technique on a scene with no emissive triangle
                               (the reference reads out of bounds there)                 */

/* SamplingTechniqueEnum.h:4-17 — values fixed by the reference's enum order. */
enum fyprt_technique {
    FYPRT_BRUTE_FORCE = 0,
    FYPRT_UNIFORM_SAMPLING = 1,
    FYPRT_COSINE_WEIGHTED_SAMPLING = 2,
    FYPRT_GGX_SAMPLING = 3,
    FYPRT_BRDF_SAMPLING = 4,
    FYPRT_LIGHT_SOURCE_SAMPLING = 5,
    FYPRT_NEE = 6,
    FYPRT_RESTIR_DI = 7,
    FYPRT_RESTIR_GI = 8
};

/* RenderingSettings.h:5-22 — field-for-field, 52 bytes, so the reference's struct can be
 * passed by address (bool == 1 byte + padding under both the MSVC x64 and SysV ABIs). */
typedef struct fyprt_settings {
    uint8_t to_accumulate;          /* toAccumulate */
    uint8_t _pad0[3];
    int32_t light_bounces;          /* lightBounces   (cast to uint8_t by the kernels, Renderer.cu:2444) */
    int32_t sample_count;           /* sampleCount    (cast to uint8_t, Renderer.cu:2480-2481)            */
    float sky_color[3];             /* skyColor */
    int32_t technique;              /* currentSamplingTechnique (enum fyprt_technique) */
    int32_t light_candidate_count;  /* lightCandidateCount */
    uint32_t rand_seed;             /* randSeed */
    uint8_t use_temporal_reuse;     /* useTemporalReuse */
    uint8_t use_spatial_reuse;      /* useSpatialReuse */
    uint8_t _pad1[2];
    int32_t temporal_history_limit; /* temporalHistoryLimit   (cast to uint8_t, Renderer.cu:1775) */
    int32_t spatial_neighbor_num;   /* spatialNeighborNum     (cast to uint8_t, Renderer.cu:1896) */
    int32_t spatial_neighbor_radius;/* spatialNeighborRadius  (cast to uint8_t, Renderer.cu:1897) */
} fyprt_settings;

/* Vertex.h:5-10 (32 B): world-space vertices, i.e. Scene::worldVertices (Scene.cpp:42-51). */
typedef struct fyprt_vertex { float position[3]; float normal[3]; float uv[2]; } fyprt_vertex;

/* Material.cuh:7-16 (44 B).  is_use_albedo_map overlays the reference's `bool` + 3 padding
 * bytes: only the low byte is read. */
typedef struct fyprt_material {
    uint32_t is_use_albedo_map;
    float albedo[3];
    uint32_t albedo_map_index;
    float roughness, metallic;
    float emission_color[3];
    float emission_power;
} fyprt_material;

/* Mesh.h:26-31 reduced to what the path reads: the triangle range of the mesh inside
 * Scene::triangles (indexStart/3, indexCount/3) and its material. */
typedef struct fyprt_mesh { uint32_t first_triangle, triangle_count; int32_t material_index; } fyprt_mesh;

/* Texture.cuh:9-14: ABGR8 pixels (A<<24|B<<16|G<<8|R), row-major. */
typedef struct fyprt_texture { const uint32_t* pixels; uint32_t width, height; } fyprt_texture;

/* LightTree.cuh:28-49 flattened (plain floats, explicit centroid because the reference's
 * UnionAABB leaves inner-node centroids at the origin, AABB.cuh:43-57). 80 bytes. */
typedef struct fyprt_lighttree_node {
    float energy;
    uint32_t num_emitters;
    uint32_t left;            /* Node::offset       (left child, inner nodes)                   */
    uint32_t right_or_emitter;/* Node::emitterIndex (right child | triangle index | mesh index) */
    uint32_t is_leaf;
    float cone_axis[3], theta_o, theta_e;   /* bounds_o */
    float box_lo[3], box_hi[3], box_centroid[3]; /* bounds_w */
    uint32_t _pad;
} fyprt_lighttree_node;

/* Optional prebuilt light trees in the reference's own shape (Scene::lightTree_tlas +
 * Mesh::lightTree_blas).  If `tlas_nodes` is NULL the library builds them itself with its
 * restatement of LightTree.cpp:21-293 (needed only by LIGHT_SOURCE_SAMPLING and NEE). */
typedef struct fyprt_lighttrees {
    const fyprt_lighttree_node* tlas_nodes; uint32_t tlas_node_count, tlas_root;
    const fyprt_lighttree_node* blas_nodes;  /* all per-mesh trees concatenated              */
    const uint32_t* blas_first;              /* [mesh_count] first node of mesh m's tree     */
    const uint32_t* blas_count;              /* [mesh_count] node count (0 = mesh not a light)*/
    const uint32_t* blas_root;               /* [mesh_count] root index relative to blas_first*/
} fyprt_lighttrees;

/* What SceneToGPU deep-copies (Scene_GPU.cpp:6-81), as flat arrays + counts. */
typedef struct fyprt_scene_desc {
    const fyprt_vertex* vertices; uint32_t vertex_count;          /* Scene::worldVertices */
    const void* triangles; uint32_t triangle_count; uint32_t triangle_stride;
        /* Scene::triangles: each record starts with {uint32 v0,v1,v2; int32 materialIndex}
           (Triangle.cuh:9-10); stride 52 passes the reference's array as is, 16 a packed one. */
    const fyprt_material* materials; uint32_t material_count;     /* Scene::materials */
    const fyprt_mesh* meshes; uint32_t mesh_count;                /* Scene::meshes    */
    const fyprt_texture* textures; uint32_t texture_count;        /* Scene::textures  */
    const uint32_t* emissive_triangles; uint32_t emissive_count;  /* Scene::emissiveTriangles; NULL = derive
                                                                     as InitSceneEmissiveTriangles, Scene.cpp:209-221 */
    const fyprt_lighttrees* light_trees;                          /* may be NULL */
} fyprt_scene_desc;

/* What CameraToGPU uploads (Camera_GPU.cu:4-60) minus the W*H ray-direction array, which is
 * regenerated on the device from inverse_projection / inverse_view with the arithmetic of
 * Camera::RecalculateRayDirections (Camera.cpp:136-153).  Matrices are column-major float[16]
 * exactly as glm::mat4 stores them. */
typedef struct fyprt_camera_desc {
    float projection[16], view[16], prev_projection[16], prev_view[16];
    float inverse_projection[16], inverse_view[16];
    float position[3];
    uint32_t viewport_width, viewport_height;
} fyprt_camera_desc;

typedef struct fyprt_context fyprt_context;

/* Per-frame statistics of the last fyprt_render call. */
typedef struct fyprt_frame_stats {
    float kernel_ms;        /* hipEvent time over the frame's kernels on the context stream */
    float kernel_ms_part[4];/* per launch (trace/shade, DI/GI part 1, part 2, ...), 0 if unused */
    uint64_t rays;          /* TraceRay invocations            } only with fyprt_set_ray_counting(ctx, 1):    */
    uint64_t box_tests;     /* AABB slab tests executed        } the SURVEY.md §8(d) instrumentation, exact    */
    uint64_t tri_tests;     /* ray/triangle tests executed     } per-lane counts reduced with device atomics   */
    uint64_t hits;          /* rays that found a triangle      }                                               */
    uint64_t part_rays[4], part_box_tests[4], part_tri_tests[4], part_hits[4];   /* the same, per launch */
    uint32_t launches;
    uint64_t node_visits;   /* 64-byte node records fetched (one per visit; box_tests counts the valid children tested in them) */
    uint64_t part_node_visits[4];
} fyprt_frame_stats;

/* Buffers readable with fyprt_read_buffer (device -> host, for parity tests). */
enum fyprt_buffer {
    FYPRT_BUF_ACCUM = 0,        /* float4 running sum          (Renderer.h:21 m_AccumulationData) */
    FYPRT_BUF_IMAGE = 1,        /* uint32 ABGR8                (Renderer.h:20 m_RenderImageData)  */
    FYPRT_BUF_PAYLOAD = 2,      /* RayHitPayload 40 B          (Renderer.h:31)                    */
    FYPRT_BUF_DEPTH = 3,        /* float                       (Renderer.h:29)                    */
    FYPRT_BUF_NORMAL = 4,       /* float2 octahedral, the frame just rendered (Renderer.h:30)     */
    FYPRT_BUF_DI_RESERVOIR = 5, /* ReSTIR_DI_Reservoir 20 B, after Part 1 (Renderer.h:34)         */
    FYPRT_BUF_DI_PREV = 6,      /* ... written by Part 2       (Renderer.h:35)                    */
    FYPRT_BUF_GI_RESERVOIR = 7, /* ReSTIR_GI_Reservoir 72 B    (Renderer.h:38)                    */
    FYPRT_BUF_GI_PREV = 8       /*                             (Renderer.h:39)                    */
};

/* ---- lifetime (the reference has none: Renderer owns raw pointers and never frees them,
 *      Renderer.cu:421-457 FreeDynamicallyAllocatedMemory is never called) */
int fyprt_create(int device_ordinal, fyprt_context** out);
void fyprt_destroy(fyprt_context* ctx);
const char* fyprt_last_error(const fyprt_context* ctx);   /* ctx may be NULL: last create error */

/* Renderer::OnResize + ResizeReservoirs/DepthBuffers/NormalBuffers/PrimaryHitPayloadBuffers
 * (Renderer.cpp:5-41, Renderer.cu:286-419): (re)allocates and zero-fills every per-pixel
 * buffer and resets the frame index to 1.  Unlike the reference the device buffers exist
 * after the FIRST call (Renderer.cpp:24-27 skips them). */
int fyprt_resize(fyprt_context* ctx, uint32_t width, uint32_t height);

/* Multi-GPU tile split (new; BASELINE.json north_star): this context renders image rows
 * [row_begin,row_end) of the full width x height frame.  ReSTIR Part 1 is additionally run
 * on `halo_rows` rows either side (halo recompute).  Default: all rows.  With spatial reuse on, halo_rows must be
 * at least spatial_neighbor_radius (or the rows must arrive by fyprt_group_* / fyprt_comm_* exchange): what a neighbour beyond
 * band + halo holds is whatever an earlier frame left there. */
int fyprt_set_rows(fyprt_context* ctx, uint32_t row_begin, uint32_t row_end, uint32_t halo_rows);
/* The interleaved split of SURVEY.md §8(e) for the techniques whose pixels are independent (every technique but the two ReSTIRs:
 * PerPixel_* of Renderer.cu:565-1626 read nothing of another pixel): the frame is cut into stripes of `stripe_rows` rows and this
 * context renders the stripes part, part + parts, part + 2 parts, ... — every part samples the whole image, so the parts cost the
 * same without balancing.  stripe_rows 0 returns to the rows of fyprt_set_rows.  A ReSTIR frame on a striped context fails with
 * FYPRT_ESTATE (spatial reuse reads the rows around a pixel).  Pixel values do not depend on the split. */
int fyprt_set_row_stripes(fyprt_context* ctx, uint32_t stripe_rows, uint32_t parts, uint32_t part);

/* SceneToGPU / FreeSceneGPU (Scene_GPU.cpp:6-163) + Renderer::SetSceneToBeUpdatedFlag(true)
 * (Renderer.h:56): builds the acceleration structure and uploads everything. */
int fyprt_upload_scene(fyprt_context* ctx, const fyprt_scene_desc* scene);

/* CameraToGPU (Camera_GPU.cu:4-60), called by Renderer::Render every frame (Renderer.cu:70). */
int fyprt_set_camera(fyprt_context* ctx, const fyprt_camera_desc* camera);

/* Renderer::Render (Renderer.cu:13-284): one frame with the given settings: selects the
 * technique's kernel(s) (Renderer.cu:87-235), accumulates, tonemaps and packs on the device,
 * then advances the frame index (Renderer.cu:258-261).  Blocking. `stats` may be NULL. */
int fyprt_render(fyprt_context* ctx, const fyprt_settings* settings, fyprt_frame_stats* stats);

/* Asynchronous variant: enqueues the frame on the context's stream and returns. */
int fyprt_render_async(fyprt_context* ctx, const fyprt_settings* settings);
int fyprt_synchronize(fyprt_context* ctx);
/* Per-launch hipEvent times (ms) of the frame enqueued `frames_back` frames ago (0 = the last one; the last 128 frames
 * are kept).  The events are recorded on the context's stream by both render variants; call after fyprt_synchronize. */
int fyprt_frame_timings(fyprt_context* ctx, uint32_t frames_back, float* kernel_ms_part4, uint32_t* launches);

/* The D2H copies at Renderer.cu:244-250: rgba8 = m_RenderImageData (ABGR8, row 0 = NDC y -1),
 * accum4 = m_AccumulationData (float4 running SUM).  Either may be NULL.  Full frame size;
 * rows outside this context's band are left untouched. */
int fyprt_readback(fyprt_context* ctx, uint32_t* rgba8, float* accum4);

/* Device pointer of the ABGR8 image (width*height uint32) for an on-device gather (RCCL). */
int fyprt_image_device_ptr(fyprt_context* ctx, void** dptr);
/* Render into a caller-owned device image buffer (e.g. a torch tensor) instead of the internal one. */
int fyprt_set_external_image(fyprt_context* ctx, void* device_ptr);
/* The context's hipStream_t (as void*) so a caller can order its own work after a frame. */
int fyprt_stream(fyprt_context* ctx, void** stream);

int fyprt_read_buffer(fyprt_context* ctx, int which /* enum fyprt_buffer */, void* dst, size_t bytes);

/* The geometry moved, the topology did not (a transform edit through SceneManager::PerformAllSceneUpdates,
 * SceneManager.cpp:24-66, where the reference rebuilds the mesh's BLAS, the TLAS and the light trees on the host): the new world
 * vertices (same count and order as uploaded) refresh the per-triangle records, the leaf triangles and the boxes of the
 * acceleration structure ON THE DEVICE; the tree keeps its shape (fyprt_upload_scene rebuilds it).  Light records are rebuilt by
 * their kernel, the light trees on the host.  Not available for scenes uploaded with prebuilt light trees. */
int fyprt_update_vertices(fyprt_context* ctx, const fyprt_vertex* vertices, uint32_t vertex_count);

/* Renderer::ResetFrameIndex / GetCurrentFrameIndex (Renderer.h:47,49). */
int fyprt_reset_frame_index(fyprt_context* ctx);
uint32_t fyprt_frame_index(const fyprt_context* ctx);

/* Export of the library's own acceleration structure as plain arrays, so an
 * instrumented CPU restatement of the same traversal can count box / triangle tests
 * (SURVEY.md §8d).  Call with NULL pointers to query the counts.
 *   nodes64: 64-byte 4-wide nodes { float origin[3]; uint8 ex[3], count; int32 child[4]; uint8 qlo[3][4], qhi[3][4]; uint32 pad[2] }
 *            child plane on axis a = origin[a] + q * 2^(ex[a]-127); child >= 0 inner node, < 0 leaf: ~child = firstTri << 2 | (n-1)
 *   tris48 : 48-byte leaf triangles { float v0[3], e1[3], e2[3]; uint32 triangleIndex; uint32 pad[2] }
 *   max_stack: worst-case number of pending traversal-stack entries of an ordered traversal (<= 31 by construction). */
int fyprt_export_bvh(fyprt_context* ctx, void* nodes64, uint32_t* node_count, void* tris48,
                     uint32_t* tri_count, int32_t* root_ref, uint32_t* max_stack);
/* Export of the light trees the library built (same flat node format as the input). */
int fyprt_export_lighttrees(fyprt_context* ctx, fyprt_lighttree_node* tlas, uint32_t* tlas_count, uint32_t* tlas_root,
                            fyprt_lighttree_node* blas, uint32_t* blas_total, uint32_t* blas_first,
                            uint32_t* blas_count, uint32_t* blas_root);

/* Count rays / box tests / triangle tests on the device (atomics; slows the frame); off by default. */
int fyprt_set_ray_counting(fyprt_context* ctx, int enabled);

/* Performance knobs (A/B experiments; defaults are the measured best).  Keys 0-7, 9-11 never change a result; keys 8 and 12
 * change the traversal order / the tree and with it only which of two triangles hit at EXACTLY the same t wins.
 * key 0: tile order — 0 linear, 1 one contiguous eighth of the tiles per XCD, 2 every 8th tile row per XCD.
 * key 1: ReSTIR DI Part 2 — 0 one thread per pixel, 1 setup kernel + shadow-task queue + persistent trace waves.
 * key 2: persistent workgroups per CU for the trace kernel (default 0 = as many as LDS and registers allow: 6 today).
 * key 3: 1 = counting-sort the shadow tasks by light bin before tracing (slotted tasks + histogram matrix + column scan +
 *        scatter, no global atomics).  Default 0: measured +0.04 ms for the sort and no faster trace — shadow-ray cost is
 *        dominated by the geometry around the ray ORIGIN, which the unsorted tile order already keeps coherent.
 * key 4: tasks a persistent wave claims per queue-head atomic (default 128).
 * key 5: idle lanes that trigger a refill of a persistent wave (default 24).
 * key 6: inner-node loop quorum of the ReSTIR DI Part-2 shadow-ray kernels and the path engine's ray kernels: lanes waiting at a leaf are
 *        served once fewer than this many lanes are still walking inner nodes (default 24; 0 = classic while-while).  The count is for a full
 *        wave; r03: a wave in which only some lanes still have a ray scales it to those lanes (rt_device.h: quorum_of).
 * key 7: the same for the kernels that trace primary rays (k_primary, k_gi_primary, ReSTIR DI Part 1, the fused small-scene frame):
 *        default 32 since r03 (bench frame 0.843 -> 0.82-0.83 ms, config 3 3.09 -> 3.03 ms; it was neutral before the node visit was trimmed); 0 = never.
 * key 8: pending-entry budget of the traversal stack rule (default 0 = a few entries above the tree's level count, chosen
 *        so that one more workgroup fits a CU's LDS; at most 31; always clamped from below to the tree's level count):
 *        siblings are pushed one by one while pending + 2 + levels(node) <= budget, else as one resume entry; the LDS stack holds budget + 1 entries per thread.
 *        Unlike keys 0-7 the value can change the visiting order (exact-t ties may resolve differently).
 * key 9: chunks every persistent wave owns statically before it starts stealing from the shared head (default 0 = auto: 2 — r03: bench frame
 *        0.833 -> 0.820 ms, config 3 3.13 -> 3.08 ms against 1; on queues shorter than the grid the static part is an even share and no atomic is issued at all).
 * key 10: smallest chunk of the guided self-scheduling of the shared part: claims shrink from key 4 towards this value as
 *        the queue runs out (default 32).
 * key 11: 1 (default) = wavefront ReSTIR DI frames are pipelined over two streams: Part 1 + setup of frame N+1 run beside the
 *        trace kernel of frame N (asynchronous frames only overlap, of course; a blocking fyprt_render waits for its frame).
 * key 12: builder of the acceleration structure for the NEXT fyprt_upload_scene: 0 (default) host binned SAH + SAH-optimal
 *        collapse; 1 device LBVH (Morton sort, Karras radix tree, collapse, refit) — milliseconds instead of a fraction of a
 *        second for a million triangles, a tree 1.65x slower to trace; 2 device PLOC (Morton sort, parallel locally-ordered
 *        clustering with search radius FYPRT_PLOC_RADIUS = 16, collapse, refit) — the same build time, a tree 1.15x slower to
 *        trace than the host builder's.  Both fall back to the host builder if the tree gets deeper than 31 wide levels.
 *        Results stay exact in every case (any valid tree finds the same closest hits but for exact-t ties).
 * key 13: MEASUREMENT ONLY (tools/band_rate.py): 1 = a lone context skips ReSTIR Part 1 on its halo rows as a band does in halo-exchange
 *        mode, without anybody filling them — the cost of one band of an exchange-mode split; the image near the band border is not valid.
 * key 14: 1 = ReSTIR DI Part-2 setup fetches every neighbour record the spatial-reuse loop can possibly visit at once (the addresses
 *        depend only on how many earlier neighbours were accepted) instead of one dependent gather per neighbour; same results.
 *        Default 0: measured slower (0.259 vs 0.225 ms) — register pressure and request rate outweigh the saved round trips.
 *        2 = the reservoir neighbourhood's hot fields (depth, normal; 12 B) of the workgroup's 76 x 76 pixel window staged in LDS (69 KB),
 *        the geometry test served from there; same results; measured slower as well (profiles/README.md r02).
 * key 15: ray kernel of the wavefront stages (techniques 0-6, ReSTIR GI): 1 = persistent waves with lane refill, 2 = one thread per ray,
 *        0 (default) = by tree size: one thread per ray below 65 536 triangles, where rays are too cheap for the refill machinery to pay.
 * key 16: EXPERIMENT, acts only in a -DRT_TOPCACHE build: the first N nodes of the (area-ordered) node array are also kept in LDS by the
 *        ReSTIR DI traversal kernels (0..1024).  Measured a net loss (profiles/README.md r03); compiled out by default.
 * key 17: techniques 0-5 on a tree of fewer than 65 536 triangles render the whole frame in ONE launch (k_path_fused: one thread per pixel,
 *        the stage path's step functions): 0 (default) = by tree size, 1 = always the stages, 2 = always fused.  Same results.
 * key 18: 1 (default) = a ReSTIR DI Part-2 shadow ray whose pixel is black in EVERY outcome (both candidate radiances exactly zero) is
 *        not traced; 0 = every Part-2 pixel traces its ray as Renderer.cu:2010-2031 does.  Same pixels; fewer rays are counted.
 *        Also on by default since r03: key 2 = 0 uses at most 4 workgroups per CU for the persistent ReSTIR DI grid while frames are pipelined.
 * key 19: ReSTIR GI Part 2.  2 (default) = ONE persistent launch (k_gi2_persistent): a lane owns a pixel of the Part-2 list for its whole neighbour
 *         loop — reservoir state in registers, visibility rays traced in place — and the lanes of a wave that have no ray in flight are serviced
 *         together (merge the verdict, next accepted neighbour, next pixel from the list), like the persistent trace kernels' refill.
 *         0 = the stages (2 x neighbours + 1 launches; state, ray and result records move through memory).  1 = one launch, one thread per
 *         pixel without refill (slower on a whole frame, kept for comparison).  Same pixels bit for bit, same ray counts in every mode.
 * key 20: k_gi2_persistent (key 19 = 2): lanes of a wave without a ray in flight before the wave services them together (default 48; 0 = default).
 * Values are range-checked (FYPRT_EINVAL): key 0: 0..2, keys 1, 3, 11, 13, 18: 0..1, keys 12, 14, 15, 17, 19: 0..2, key 2: 0..16, keys 5, 6, 7, 20: 0..64,
 * key 8: 0..31, key 16: 0..1024; keys 19..23 are reserved (0). */
int fyprt_set_tuning(fyprt_context* ctx, int key, int value);
/* The value in effect (key 8: the budget actually used for the uploaded scene, which an instrumented restatement of the
 * traversal must use too). */
int fyprt_get_tuning(fyprt_context* ctx, int key, int* value);

/* Scene::vertices (object space) + every mesh's vertex range [mesh_first_vertex[m], mesh_first_vertex[m+1]) (Mesh::vertexStart /
 * vertexCount), kept on the device so that fyprt_update_transforms can apply a transform edit there.  After fyprt_upload_scene. */
int fyprt_set_object_vertices(fyprt_context* ctx, const fyprt_vertex* object_vertices, uint32_t vertex_count, const uint32_t* mesh_first_vertex);
/* A transform edit of `count` meshes — SceneManager::PerformAllSceneUpdates with meshTransformToBeUpdated (SceneManager.cpp:24-66),
 * i.e. Mesh::worldTransformMatrix (column-major glm::mat4, 16 floats per listed mesh) applied to the mesh's object-space vertices as
 * Scene.cpp:42-51 does (position / w; normal by the model matrix with w = 0, normalised).  64 bytes per mesh cross the bus; world
 * vertices, per-triangle records, tree boxes and light records are refreshed on the device, the light trees of the moved emissive
 * meshes (+ the TLAS) on the host.  Same result as fyprt_update_vertices with the host-computed world vertices. */
int fyprt_update_transforms(fyprt_context* ctx, const uint32_t* mesh_indices, const float* matrices16, uint32_t count);

/* MisUtils::ComputeMSE / ComputePSNR (MisUtils.cpp:118-157) of the current frame against a host reference image, reduced on the
 * device (exact integer sums: equals the host routine bit for bit); `flip_reference_rows` reads the reference vertically flipped as
 * ComputeMSE reads its BMP original.  The benchmark workflow of WalnutApp.cpp:826-876 without a read-back.  `psnr` may be NULL. */
int fyprt_compare_image(fyprt_context* ctx, const uint32_t* reference_rgba8, int flip_reference_rows, double* mse, double* psnr);
/* Self-test of the arithmetic contract: the library's short correctly rounded sqrt / 1/x / 1/sqrt(x) sequences (rt_math.h) against the
 * compiler's IEEE sequences on ALL 2^32 binary32 arguments each.  mismatches[3] (and, optionally, the smallest offending argument's bits)
 * in the order sqrt, reciprocal, reciprocal square root; all zero on a sound build.  New (no reference counterpart). */
int fyprt_selftest_math(fyprt_context* ctx, uint64_t* mismatches3, uint32_t* first_bad3 /* may be NULL */);

/* ================================================================================================= multi-GPU
 * The reference renders on one GPU (Renderer.cu:13-284); there is no reference interface for this section.  It splits ONE
 * Renderer::Render call over several GPUs by image rows: every GPU holds the whole scene and renders a band;
 * reservoirs, G-buffers and accumulation stay on the GPU that owns the rows; the RGBA8 bands are gathered once per frame.
 * `row_bounds` always has (number of bands + 1) entries: band k = rows [row_bounds[k], row_bounds[k+1]), row_bounds[0] = 0,
 * the last entry = height.
 * Halo mode (the `radius` rows either side of a band that ReSTIR Part 2's spatial reuse reads):
 *   0 recompute: every band also runs Part 1 on its halo rows; frame 1 equals the single-GPU frame, later frames differ (unbiased)
 *                near band borders because the halo rows have no temporal history;
 *   1 exchange : the bands send each other the Part-1 records (and the temporal history) of those rows — a static-camera
 *                sequence then equals the single-GPU sequence bit for bit on every frame, and Part 1 does no duplicate work
 *                (shown for the peer-copy transport fyprt_group_*; the RCCL transport fyprt_comm_* runs the same plan but has only
 *                executed with a one-rank communicator so far: INTEGRATION.md, "Verification status"). */
typedef struct fyprt_group fyprt_group;
/* --- one process, one context per GPU (a C++ host such as the reference's MainLayer): peer copies, no collective library */
int fyprt_group_create(fyprt_context** contexts, int n, const uint32_t* row_bounds, fyprt_group** out);
void fyprt_group_destroy(fyprt_group* group);                       /* the contexts stay alive */
int fyprt_group_set_rows(fyprt_group* group, const uint32_t* row_bounds);
int fyprt_group_set_halo_mode(fyprt_group* group, int mode);
/* stripe_rows > 0: frames of the per-pixel techniques are split into interleaved stripes (fyprt_set_row_stripes, context i = part i),
 * ReSTIR frames keep the row bands; the gather moves stripes instead of bands.  The rows a context accumulates are the rows it
 * renders: restart the accumulation (fyprt_reset_frame_index on every context) when changing this setting, and — while it is on —
 * when switching between a ReSTIR and a per-pixel technique (the reference's host restarts it on any change of settings anyway). */
int fyprt_group_set_interleave(fyprt_group* group, uint32_t stripe_rows);
int fyprt_group_render(fyprt_group* group, const fyprt_settings* settings);   /* one frame on every band; asynchronous */
int fyprt_group_gather(fyprt_group* group, int root);               /* all bands' RGBA8 rows into context `root`'s image; asynchronous */
int fyprt_group_synchronize(fyprt_group* group);
/* --- one process per GPU: RCCL over xGMI (librccl.so.1 is opened on first use).  Rank 0 calls fyprt_comm_unique_id and hands
 *     the 128 bytes to the other ranks (any transport); every rank then calls fyprt_comm_init_rank on its resized context. */
int fyprt_comm_unique_id(void* id128);
int fyprt_comm_init_rank(fyprt_context* ctx, int world_size, int rank, const void* id128, const uint32_t* row_bounds);
int fyprt_comm_set_rows(fyprt_context* ctx, const uint32_t* row_bounds);
int fyprt_comm_set_halo_mode(fyprt_context* ctx, int mode);
int fyprt_comm_set_interleave(fyprt_context* ctx, uint32_t stripe_rows);       /* as fyprt_group_set_interleave; same value on every rank */
int fyprt_comm_render(fyprt_context* ctx, const fyprt_settings* settings);    /* this rank's band; collective (every rank calls it); asynchronous */
int fyprt_comm_gather(fyprt_context* ctx, int root /* < 0: every rank gets the frame */);   /* grouped ncclBroadcast per band, in place */
void fyprt_comm_destroy(fyprt_context* ctx);
/* --- building blocks */
/* The two parts of a ReSTIR frame as separate asynchronous calls, for a host with its own transport for the halo rows. */
int fyprt_render_part(fyprt_context* ctx, const fyprt_settings* settings, int part /* 1 or 2 */);
/* Cost-balanced bands: new boundaries from the milliseconds each band took (fyprt_last_frame_ms), at most `max_shift` rows per
 * boundary and call, bands at least `min_rows` high.  Pure arithmetic: the rows change owner with fyprt_group_set_rows /
 * fyprt_comm_set_rows, which move their accumulation and temporal history along. */
int fyprt_balance_rows(const uint32_t* row_bounds, const float* band_ms, int n, uint32_t min_rows, uint32_t max_shift, uint32_t* new_bounds);
int fyprt_last_frame_ms(fyprt_context* ctx, float* ms);
/* The transfers of one halo exchange: (receiver, owner, first row, end row) per entry; returns the number of entries. */
int fyprt_halo_plan(const uint32_t* row_bounds, int n, uint32_t halo, uint32_t height, int wrap_row, uint32_t* out4, int capacity);
/* The point-to-point operations rank `rank` issues inside ONE RCCL group section, in issue order — kind 0: a halo exchange over
 * `row_bounds`; kind 1: fyprt_comm_set_rows from `row_bounds` to `new_bounds`.  (is_recv, peer, buffer, byte offset, bytes) per
 * operation; returns their number.  Pure host arithmetic (no device, no RCCL): lets a test check that the two ends of every pair of
 * ranks list the same byte counts in the same order, which is what RCCL's matching needs. */
int fyprt_comm_ops(int kind, const uint32_t* row_bounds, const uint32_t* new_bounds, int n, uint32_t halo, uint32_t height, int wrap_row, uint32_t width,
                   int rank, const uint32_t* bytes_per_pixel, int nbuf, uint64_t* out5, int capacity);

/* Library / build identification ("fyprt <version> gfx950 ..."). */
const char* fyprt_version(void);

#ifdef __cplusplus
}
#endif
#endif /* FYPRT_H */
