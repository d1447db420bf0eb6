/*
 * rt_hip.h — C ABI of the MI355X (gfx950) backend for the per-sample trace loop of
 * Ararem.RayTracer's RenderJob.
 *
 * The reference has no FFI of its own: the boundary this library replaces is the body of
 * RenderJob.RenderInternal (Ararem.RayTracer.Core/RenderJob.cs:69-126) and everything it calls.
 * A C# backend (csharp/HipRenderBackend.cs) flattens the Scene graph into
 * the plain-old-data tables below and drives the entry points through P/Invoke.
 *
 * Conventions
 *   - every entry point returns 0 on success and a negative rt_status on failure;
 *     rt_last_error() returns a thread-local, NUL-terminated UTF-8 description of the last failure.
 *   - the caller owns every host pointer; the library copies what it needs during the call.
 *   - one handle => one calling thread at a time; different handles are independent.
 *   - all floating-point is IEEE binary32, vectors are x,y,z, matrices follow System.Numerics
 *     (row-vector convention: v' = v * M, element Mrc = row r, column c).
 *   - an object's id is its index in rt_scene_desc.objects (== index in Scene.SceneObjects);
 *     a material's id is its index in rt_scene_desc.materials (one entry per distinct Material
 *     *reference*: RefractiveMaterial.cs:51-52 and EmissiveRefractiveMaterial.cs:53-63 compare
 *     materials by identity).
 *   - user-defined Hittable/Material/Texture/Light/SkyBox subclasses have no table here: the
 *     flattener must reject them (no CPU fallback exists in this library).
 */
#ifndef RT_HIP_H
#define RT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_ABI_VERSION 5

typedef struct rt_scene  rt_scene;   /* opaque: flattened Scene + BVH on one device */
typedef struct rt_render rt_render;  /* opaque: one RenderJob's device state */

typedef enum rt_status {
    RT_OK                = 0,
    RT_ERR_INVALID_ARG   = -1, /* null pointer, bad enum, inconsistent table sizes */
    RT_ERR_UNSUPPORTED   = -2, /* type the kernels do not implement */
    RT_ERR_NO_DEVICE     = -3, /* no gfx950 device visible, or HIP runtime missing */
    RT_ERR_HIP           = -4, /* a HIP runtime call failed; see rt_last_error() */
    RT_ERR_OUT_OF_MEMORY = -5,
    RT_ERR_TOO_LARGE     = -6  /* size beyond a documented limit */
} rt_status;

/* ------------------------------------------------------------------------------------------
 * Hittables  (Ararem.RayTracer.Impl/Hittables).  Each record holds exactly the fields the
 * reference's TryHit/FastTryHit read, including the values its constructor caches, so that a
 * flattener inside the C# class can copy private fields verbatim.
 * ------------------------------------------------------------------------------------------ */
typedef enum rt_hittable_type {
    RT_HIT_SPHERE         = 0,  /* Sphere.cs */
    RT_HIT_QUAD           = 1,  /* Quad.cs */
    RT_HIT_TRIANGLE       = 2,  /* NEW (no reference counterpart; defined by analogy with Quad, SURVEY §8a row aT) */
    RT_HIT_INFINITE_PLANE = 3,  /* InfinitePlane.cs */
    RT_HIT_XY_PLANE       = 4,  /* XYPlane.cs */
    RT_HIT_XZ_PLANE       = 5,  /* XZPlane.cs */
    RT_HIT_YZ_PLANE       = 6,  /* YZPlane.cs */
    RT_HIT_DISK           = 7,  /* Disk.cs */
    RT_HIT_BOX            = 8,  /* Box.cs */
    RT_HIT_CAPSULE        = 9,  /* Capsule.cs */
    RT_HIT_CYLINDER       = 10, /* Cylinder.cs */
    RT_HIT_MEDIUM         = 11, /* ConstantDensityMedium.cs */
    RT_HIT_TYPE_COUNT     = 12
} rt_hittable_type;

typedef struct rt_sphere {           /* Sphere.cs:17-22 (radiusSqr = radius*radius is recomputed) */
    float centre[3];
    float radius;
} rt_sphere;

typedef struct rt_quad {             /* Quad.cs:37-55 */
    float origin[3];
    float normal[3];                 /* Normalize(Cross(u, v)) */
    float inv_col1[4];               /* localToQuadMatrix M11, M21, M31, M41 (Quad.cs:107) */
    float inv_col2[4];               /* localToQuadMatrix M12, M22, M32, M42 (Quad.cs:108) */
} rt_quad;

typedef struct rt_triangle {         /* NEW primitive, see rt_hittable_type */
    float v0[3];
    float e1[3];                     /* v1 - v0 */
    float e2[3];                     /* v2 - v0 */
    float normal[3];                 /* Normalize(Cross(e1, e2)), never flipped (as Quad) */
    float inv_col1[4];               /* same construction as rt_quad with u=e1, v=e2 */
    float inv_col2[4];
} rt_triangle;

typedef struct rt_infinite_plane {   /* InfinitePlane.cs:12-20; normal is NOT normalised by the ctor */
    float point[3];
    float normal[3];
    float neg_point_dot_normal;
} rt_infinite_plane;

typedef struct rt_axis_plane {       /* XYPlane.cs:19-29 (a=x,b=y,c=z) / XZPlane.cs:19-29 (a=x,b=z,c=y) / YZPlane.cs:17-26 (a=y,b=z,c=x) */
    float a_lo, a_hi;
    float b_lo, b_hi;
    float c;
} rt_axis_plane;

typedef struct rt_disk {             /* Disk.cs:23-41 */
    float centre[3];
    float normal[3];                 /* normalised by the ctor */
    float radius;
    float neg_centre_dot_norm;
    float inv_col1[4];               /* localToDiskMatrix M11, M21, M31, M41 */
    float inv_col2[4];               /* localToDiskMatrix M12, M22, M32, M42 */
} rt_disk;

typedef struct rt_box {              /* Box.cs:35-65 */
    float world_to_box[12];          /* WorldToBoxTransform rows 1..4, columns 1..3: M11 M12 M13 M21 ... M43 */
    float box_to_world_rows[9];      /* boxToWorldRow1, Row2, Row3 (Box.cs:62-64) */
} rt_box;

typedef struct rt_capsule {          /* Capsule.cs:35-55 */
    float p1[3];
    float p2[3];
    float radius;
    float uv_matrix[9];              /* uvMatrix M11 M12 M13 M21 M22 M23 M31 M32 M33 (Capsule.cs:48) */
} rt_capsule;

typedef struct rt_cylinder {         /* Cylinder.cs:28-40 */
    float p1[3];
    float p2[3];
    float radius;
} rt_cylinder;

typedef struct rt_medium {           /* ConstantDensityMedium.cs:23-30 */
    uint32_t boundary_type;          /* rt_hittable_type of Boundary (not RT_HIT_MEDIUM) */
    uint32_t boundary_index;         /* index into that type's table */
    float    density;
    float    neg_inv_density;        /* -1f / density */
} rt_medium;

/* One entry per Scene.SceneObjects element, in order (SceneObject.cs:6-8).
 * aabb = Hittable.BoundingVolume exactly as the reference computes it (may be +-infinity,
 * InfinitePlane.cs:28; such objects are kept outside the BVH). */
typedef struct rt_object {
    uint32_t type;                   /* rt_hittable_type */
    uint32_t index;                  /* index into the table of that type */
    uint32_t material;               /* index into materials (the Hittable's Material) */
    float    aabb_min[3];
    float    aabb_max[3];
} rt_object;

/* ------------------------------------------------------------------------------------------
 * Textures (Impl/Textures), materials (Impl/Materials), lights (Impl/Lights), skyboxes.
 * ------------------------------------------------------------------------------------------ */
typedef enum rt_texture_type {
    RT_TEX_SOLID  = 0,               /* SolidColourTexture.cs:14 : f[0..2] = colour */
    RT_TEX_MARBLE = 1,               /* MarbleTexture.cs:18-48  : f[0]=Scale f[1]=NoiseScale f[2]=NoiseStrength f[3]=DropoffPower */
    RT_TEX_GREYSCALE_PERLIN = 2      /* GreyscaleNoiseTexture.cs:20 whose Module is a SharpNoise Perlin (any other Module is opaque: rejected):
                                        f[0]=Frequency f[1]=Lacunarity f[2]=Persistence f[3]=OctaveCount f[4]=Seed (integers as floats);
                                        colour = (v, v, v), v = the module's value at WorldPoint, NOT clamped (GreyscaleNoiseTexture.cs:20).
                                        Same stand-in gradient noise as MarbleTexture: parity unpinned (SharpNoise is not in the tree) */
} rt_texture_type;

typedef struct rt_texture {
    uint32_t type;
    float    f[7];
} rt_texture;

typedef enum rt_material_type {
    RT_MAT_STANDARD            = 0,  /* StandardMaterial.cs  : tex[0]=Albedo tex[1]=Emission f[0]=Diffusion */
    RT_MAT_REFRACTIVE          = 1,  /* RefractiveMaterial.cs: tex[0]=Tint f[0]=RefractiveIndex flags bit0=AlternateRefractionMode */
    RT_MAT_EMISSIVE_REFRACTIVE = 2,  /* EmissiveRefractiveMaterial.cs: as above + tex[1]=Emission, flags bit1=DirectEmission */
    RT_MAT_PHONG               = 3,  /* PhongMaterial.cs     : f[0..2]=Specular f[3..5]=Diffuse f[6..8]=Ambient f[9]=Shininess */
    RT_MAT_VOLUMETRIC          = 4   /* ConstantDensityMedium.cs:87-130 : f[0..2]=Albedo f[3]=Density */
} rt_material_type;

#define RT_MAT_FLAG_ALTERNATE_REFRACTION 1u
#define RT_MAT_FLAG_DIRECT_EMISSION      2u

typedef struct rt_material {
    uint32_t type;
    uint32_t tex[2];
    uint32_t flags;
    float    f[10];
} rt_material;

typedef enum rt_light_type {
    RT_LIGHT_POINT          = 0,     /* PointLight.cs:9 */
    RT_LIGHT_DIFFUSE_SPHERE = 1,     /* DiffuseSphereLight.cs:88-142 (the hard-wired `case 4`) */
    RT_LIGHT_DIFFUSE_SHAPED = 2      /* DiffuseShapedLight.cs:69-105 */
} rt_light_type;

/* SimpleLightBase.DistanceAttenuationFunc is a delegate (SimpleLightBase.cs:132-139; factories :83-129); the flattener
 * classifies it into one of the factory families below or rejects the light. x = kMax / AttenuationRadius. */
typedef enum rt_attenuation_kind {
    RT_ATTEN_CONSTANT        = 0,    /* p[0]                        (e.g. BuiltinScenes.cs:146 `(_, _) => 1f`) */
    RT_ATTEN_LINEAR          = 1,    /* 1 - x                        SimpleLightBase.cs:83 */
    RT_ATTEN_POWER           = 2,    /* Pow(1 - x, p[0])             :95-103 */
    RT_ATTEN_POWER_STAY_HIGH = 3,    /* 1 - Pow(x, p[0])             :95-103 (stayHighInitially) */
    RT_ATTEN_LOGISTIC        = 4,    /* 1 / (1 + Pow(E, p[1] * (x - p[0])))   midpoint p[0], steepness p[1]  :115 */
    RT_ATTEN_LOGISTIC_LKX    = 5,    /* p[0] / (1 + Pow(E, -p[1] * (x - p[2])))                              :125 */
    RT_ATTEN_EXP_DECAY       = 6     /* Pow(E, p[0] * -x)            :129 (default, p[0]=5, :132) */
} rt_attenuation_kind;

typedef struct rt_light {
    uint32_t type;
    float    position[3];
    float    colour[3];
    float    attenuation_radius;     /* SimpleLightBase.cs:27 */
    float    cutoff_radius;          /* :34 */
    float    diffusion_radius;       /* DiffuseSphereLight.cs:13 */
    uint32_t atten_kind;
    float    atten_p[3];
    uint32_t shape_type;             /* DIFFUSE_SHAPED: rt_hittable_type of Shape ... */
    uint32_t shape_index;            /* ... and its index in that table (need not be a scene object) */
    float    shape_aabb_min[3];      /* Shape.BoundingVolume */
    float    shape_aabb_max[3];
} rt_light;

typedef enum rt_skybox_type {
    RT_SKY_DEFAULT       = 0,        /* DefaultSkyBox.cs:10-14 */
    RT_SKY_SINGLE_COLOUR = 1         /* SingleColourSkyBox.cs:16 */
} rt_skybox_type;

typedef struct rt_skybox {
    uint32_t type;
    float    colour[3];
} rt_skybox;

typedef struct rt_camera {           /* the fields Camera.GetRay reads, Camera.cs:35-42 */
    float look_from[3];
    float horizontal[3];
    float vertical[3];
    float lower_left_corner[3];
    float u[3];
    float v[3];
    float lens_radius;
} rt_camera;

typedef struct rt_scene_desc {
    uint32_t struct_size;            /* sizeof(rt_scene_desc) */
    uint32_t abi_version;            /* RT_ABI_VERSION */
    rt_camera camera;
    rt_skybox skybox;
    uint32_t n_objects;         const rt_object*         objects;
    uint32_t n_spheres;         const rt_sphere*         spheres;
    uint32_t n_quads;           const rt_quad*           quads;
    uint32_t n_triangles;       const rt_triangle*       triangles;
    uint32_t n_infinite_planes; const rt_infinite_plane* infinite_planes;
    uint32_t n_xy_planes;       const rt_axis_plane*     xy_planes;
    uint32_t n_xz_planes;       const rt_axis_plane*     xz_planes;
    uint32_t n_yz_planes;       const rt_axis_plane*     yz_planes;
    uint32_t n_disks;           const rt_disk*           disks;
    uint32_t n_boxes;           const rt_box*            boxes;
    uint32_t n_capsules;        const rt_capsule*        capsules;
    uint32_t n_cylinders;       const rt_cylinder*       cylinders;
    uint32_t n_media;           const rt_medium*         media;
    uint32_t n_materials;       const rt_material*       materials;        /* at most 65 534 (RT_ERR_TOO_LARGE beyond): the path state keeps material ids in 16 bits */
    uint32_t n_textures;        const rt_texture*        textures;
    uint32_t n_lights;          const rt_light*          lights;
} rt_scene_desc;

/* RenderOptions (Ararem.RayTracer.Core/RenderOptions.cs:11-19) minus what stays host-side
 * (ConcurrencyLevel, Passes/InfinitePasses: the caller loops over rt_render_passes). */
typedef enum rt_debug_vis {          /* GraphicsDebugVisualisation.cs:4-61, same values; RenderJob.cs:148-309 says what each one shows */
    RT_VIS_NONE = 0,
    RT_VIS_NORMALS = 1,              /* (N + 1) / 2 of the first hit                                            :156-162 */
    RT_VIS_FACE_DIRECTION = 2,       /* OutsideFace ? green : red                                               :163-168 */
    RT_VIS_DISTANCE_CLOSE = 3,       /* Pow(E, -(0.01 z^2)), z = K - KMin                                       :181-188 */
    RT_VIS_DISTANCE_MID = 4,         /* Pow(E, -(0.0001 z^2))                                                   :189-196 */
    RT_VIS_DISTANCE_FAR = 5,         /* Pow(E, -(0.000001 z^2))                                                 :197-204 */
    RT_VIS_UV_COORDS = 6,            /* (u, v, 1)                                                               :229-233 */
    RT_VIS_PIXEL_COORD_TEXTURE = 7,  /* per-material colour, checker from Sin(x/2) Sin(y/2) < 0                 :206-210 */
    RT_VIS_SCATTER_DIRECTION = 8,    /* (Scatter(hit).Direction + 1) / 2                                        :221-228 */
    RT_VIS_LOCAL_COORD_TEXTURE = 9,  /* checker from Sin(40 lx) Sin(40 ly) Sin(40 lz) < 0 on HitRecord.LocalPoint :216-220 */
    RT_VIS_WORLD_COORD_TEXTURE = 10, /* the same on WorldPoint                                                  :211-215 */
    RT_VIS_ESTIMATED_LIGHT = 11,     /* sum of Light.CalculateLight over lights and samples / samples           :234-250 */
    RT_VIS_UNDEFINED_TEST = 12,      /* the reference throws ArgumentOutOfRangeException (:307-308): RT_ERR_INVALID_ARG */
    RT_VIS_BOUNCE_DEPTH = 13         /* bounce loop without colours; Lerp(White, Blue*0.02, 1 - Pow(E, -(d^2 Sqrt(MaxBounceDepth)))), purple on a primary miss :251-306 */
} rt_debug_vis;
/* The per-material colour of the three "debug textures" is UniqueColourFromMaterialHash (RenderJob.cs:317-328): bytes [o..o+2] of
 * Material.GetHashCode(), o = 1 for the alternate checker cell.  GetHashCode of a reference type is a per-process identity hash, so it
 * cannot be reproduced; this library uses hash = (material id + 1) * 2654435761 (mod 2^32) in its place. */

#define RT_MAX_DEVICES 16
/* rt_render_opts.flags */
#define RT_RENDER_EXACT_LIGHT_SAMPLING 1u   /* shadow rays and light weights in the reference's correctly rounded arithmetic (Normalize, Distance,
                                               kMax / AttenuationRadius, MathF.Pow(MathF.E, y), sum / samples) instead of the hardware's 1-ulp
                                               rsq / rcp / exp2: bit-identical shadow rays to a CPU restatement, ~10 % slower shadow kernel.
                                               The environment variable RT_HIP_EXACT_LIGHTS=1 sets it for every render. */
#define RT_RENDER_EXACT_OCCLUSION 2u        /* every shadow ray answers exactly as the reference's scan of all objects (BinaryBvhNode.cs:41,
                                               AsyncRenderJob AnyIntersectionFast), even where that answer is binary32 noise: a sphere test
                                               at distance D errs by ~1e-3 D, so from a ground point 10 000 units out a "hit" can lie 10 units
                                               outside the sphere's box, and the box tests of such a ray must let the whole scene through.
                                               Without the flag the boxes of a shadow ray grow by at most half the smallest round radius
                                               (exact up to a reach of ~1 100 smallest radii); only pixels whose shadow rays start farther than that
                                               from the objects can differ (a ground plane seen towards the horizon), and a dense scene renders up to 20x faster.  rt_trace_any is always exact.
                                               RT_HIP_EXACT_OCCLUSION=1 sets it for every render. */
#define RT_RENDER_VALIDATE 4u               /* GraphicsValidator in full (RenderJob.cs:588-607, GraphicsValidator.cs:35-45): every closest hit of an
                                               ordinary render gets its UV computed and checked as well as its normal, and both warn-only rules are
                                               counted PER OBJECT (rt_render_validator_counts), as GraphicsValidator.Errors files them.  Off by
                                               default: the UVs cost the shading kernel an atan2 / acos per sphere hit and the scene-specific
                                               kernel variant.  Scenes of at most 65 536 objects (RT_ERR_TOO_LARGE beyond).  The image does not change. */

typedef struct rt_render_opts {
    uint32_t struct_size;            /* sizeof(rt_render_opts) */
    uint32_t width, height;          /* RenderWidth, RenderHeight (>= 1) */
    float    k_min, k_max;           /* KMin (0.001), KMax (+inf) */
    uint32_t max_bounce_depth;       /* MaxBounceDepth (10) */
    uint32_t light_sample_count_hint;/* LightSampleCountHint (2), >= 1 */
    uint32_t debug_visualisation;    /* rt_debug_vis */
    uint64_t seed;                   /* key of the counter RNG that replaces Random.Shared (RandUtils.cs:17) */
    uint32_t passes_per_batch;       /* passes traced per wavefront batch; 0 = library default (see max_passes_per_call; a multiple of 8 where there is room: a batch's passes are traced in groups) */
    int32_t  device;                 /* HIP device ordinal, -1 = the scene's device (ignored when n_devices > 1) */
    void*    stream;                 /* hipStream_t to enqueue on, NULL = a stream owned by the library (must be NULL when n_devices > 1) */
    float*   accum_device;           /* optional caller-owned DEVICE buffer of width*height*3 floats (zeroed by the
                                        caller) that receives the colour sums, e.g. a buffer a collective reduces;
                                        NULL = owned by the library (must be NULL when n_devices > 1) */
    /* ---- ABI 2 ---- */
    uint32_t flags;                  /* RT_RENDER_* */
    uint32_t max_passes_per_call;    /* the largest n_passes this caller gives rt_render_passes, 0 = unknown.  The default batch holds
                                        min(64, 128 M paths / pixels, this) passes: a GUI that renders 8 passes per call does not pay
                                        for 64 passes of path state (184 B per path and batch slot) */
    uint32_t n_devices;              /* 0 or 1: one GPU, the scene's.  G > 1: the passes of every rt_render_passes call are sharded over
                                        device_ids[0..G) inside the library (pass first + i*stride goes to device i mod G; the scene is
                                        replicated by peer copies, one host thread and stream set per device) and the readers
                                        (rt_render_read_rgb24 / _accum / _stats) combine the shards on device_ids[0] in device order.
                                        An ordinal may repeat (two shards on one GPU: how the 1-GPU test box exercises this path). */
    int32_t  device_ids[RT_MAX_DEVICES];
} rt_render_opts;

/* RenderStats (Ararem.RayTracer.Core/RenderStats.cs:12-93), field for field, plus counters the
 * reference does not keep (shadow rays are not counted by RenderJob.cs:575). */
#define RT_DEPTH_HIST_SIZE 64
typedef struct rt_stats {
    uint64_t pixels_rendered;        /* RenderJob.cs:108 */
    uint64_t passes_rendered;        /* :120 */
    uint64_t material_scatter_count; /* :439 */
    uint64_t material_absorbed_count;/* :431 */
    uint64_t aabb_misses;            /* BinaryBvhNode.cs:18, SingleObjectBvhNode.cs:15: box tests of the closest-hit traversal that failed.  The device tree has
                                        another topology than the reference's (BvhTree.cs:181-186 is implementation-defined), so the count differs from the
                                        reference's for the same scene; aabb_misses <= node_visits */
    uint64_t hittable_misses;        /* SingleObjectBvhNode.cs:28: Hittable.TryHit calls that returned null; hittable_misses + hittable_intersections == prim_tests */
    uint64_t hittable_intersections; /* SingleObjectBvhNode.cs:23: Hittable.TryHit calls that returned a hit */
    uint64_t sky_rays;               /* RenderJob.cs:447 */
    uint64_t bounce_limit_exceeded;  /* :418 */
    uint64_t ray_count;              /* :583  closest-hit rays */
    uint64_t shadow_rays;            /* AnyIntersectionFast calls (:575) */
    uint64_t node_visits;            /* V: box tests during closest-hit traversal */
    uint64_t prim_tests;             /* T: primitive tests during closest-hit traversal */
    uint64_t shadow_node_visits;     /* V for any-hit traversal */
    uint64_t shadow_prim_tests;      /* T for any-hit traversal */
    uint64_t normals_wrong_magnitude;/* closest hits whose normal fails CheckVectorNormalized (GraphicsValidator.cs:52-60,
                                        |n.n - 1| > 0.01; RenderJob.cs:588-598 logs and counts it, changes nothing) — e.g. an InfinitePlane built with a normal
                                        that is not a unit vector (InfinitePlane.cs:20-25 keeps it as given).  Summed over objects. */
    uint64_t uv_invalid;             /* CheckUVCoordValid fails (GraphicsValidator.cs:62-66; RenderJob.cs:600-607 records GraphicsErrorType.UVInvalid): closest hits with a UV outside [0, 1]^2.
                                        Counted where the device computes UVs: debug-visualisation renders and renders with RT_RENDER_VALIDATE (every hit); 0 otherwise */
    uint64_t ray_depth_counts[RT_DEPTH_HIST_SIZE]; /* RenderStats.RayDepthCounts (:458); depths >= 63 land in [63] */
} rt_stats;

typedef struct rt_ray {              /* Ray.cs:14-26; direction is used as given (callers normalise) */
    float origin[3];
    float direction[3];
} rt_ray;

typedef struct rt_hit {              /* the HitRecord fields parity checks compare (HitRecord.cs:19-20) */
    float    k;                      /* HitRecord.K; NaN-free; undefined when object == RT_NO_HIT */
    uint32_t object;                 /* index into objects, RT_NO_HIT on a miss */
    uint32_t material;
    float    normal[3];
    float    uv[2];
    float    world_point[3];
    uint32_t outside_face;
} rt_hit;
#define RT_NO_HIT 0xFFFFFFFFu

/* ---------------- entry points ---------------- */

int rt_version(void);                               /* RT_ABI_VERSION of the built library */
const char* rt_last_error(void);
int rt_device_count(int* n);

/* Flattened Scene (Scene.cs:9) -> device tables + BVH (replaces BvhTree ctor, BvhTree.cs:22-136). */
int rt_scene_create(const rt_scene_desc* desc, int device, rt_scene** out);
int rt_scene_destroy(rt_scene* scene);

/* What the library built for a scene: the shape of its BVH (the reference exposes RenderJob.BvhTree, RenderJob.cs:660) and
 * which kernel variant / tree placement the trace kernels will use. */
typedef struct rt_scene_info {
    uint32_t n_objects, n_unbounded;  /* objects outside the BVH because their BoundingVolume is not finite */
    uint32_t n_nodes, bvh_depth;      /* inner nodes (64 B each) and deepest inner level */
    uint32_t tree_in_lds;             /* 1: the tree and sphere table are copied into each workgroup's LDS */
    uint32_t lds_image_bytes;
    uint32_t built_on_device;         /* 1: built on the GPU (binned SAH, the default from 256 k bounded objects; RT_HIP_BVH=sah / =ploc / =device choose a device builder for any size), 0: SAH on the host (RT_HIP_BVH=host forces it) */
    uint32_t object_list;             /* 1: <= 16 bounded objects, traced as a list (every object for every ray, as the reference's any-hit does) instead of the tree */
    char     variant[16];             /* kernel variant specialised on the scene's content: "spheres", "mesh", "boxes", "generic" */
    uint32_t wide_nodes, wide_levels; /* trees read through the caches (>= 1024 inner nodes; RT_HIP_WIDE=0/1 forces either): the 8-wide tree with quantised inner boxes made from the binary one — its node count and depth, 0 when the binary tree is traced */
    int32_t  device;                  /* the HIP device the scene's tables live on (rt_scene_create's `device`, or HIP's current device then) */
    uint32_t reserved;
} rt_scene_info;
int rt_scene_get_info(rt_scene* scene, rt_scene_info* out);

/* RenderJob ctor (RenderJob.cs:34-62): allocates the colour-sum and sample-count state. */
int rt_render_create(rt_scene* scene, const rt_render_opts* opts, rt_render** out);
int rt_render_destroy(rt_render* render);

/* The pass loop body of RenderInternal (RenderJob.cs:69-126), :92-122, for passes first_pass, first_pass+pass_stride, ...
 * (n_passes of them). pass_stride > 1 is how a caller shards passes over devices (pass p goes to
 * device p mod G). Blocking. */
int rt_render_passes(rt_render* render, uint64_t first_pass, uint32_t n_passes, uint32_t pass_stride);

/* The resolve at the end of UpdateBuffers (RenderJob.cs:626-655), :646-654: avg -> Clamp01 -> Sqrt -> (byte)(255*c); row 0 = top.
 * Before the first pass the image is black, as the reference's freshly constructed Image<Rgb24> is. */
int rt_render_read_rgb24(rt_render* render, uint8_t* dst /* width*height*3 */);
/* rawColourBuffer / sampleCountBuffer (RenderJob.cs:669-672), image-row order (row 0 = top). */
int rt_render_read_accum(rt_render* render, float* rgb_sum /* width*height*3 */, uint32_t* sample_count /* 1 value */);
/* Resolve an externally combined sum (e.g. after an RCCL reduce into accum_device) with the given count. */
int rt_render_resolve_device(rt_render* render, const float* accum_device, uint32_t sample_count, uint8_t* dst_host);
int rt_render_stats(rt_render* render, rt_stats* out);

/* Per-kernel timing of the trace loop, measured with HIP events on the stream the kernels are launched on.
 * No reference counterpart (the reference only keeps a Stopwatch, RenderJob.cs:72,124); bench.py's roofline
 * figures come from here.  Off by default; while on, every launch is bracketed by two event records.
 * (RT_K_RAYGEN stays 0 in an ordinary render: the first closest-hit and shading launches make the camera rays themselves;
 * only debug visualisations store them with a kernel of their own.) */
typedef enum rt_kernel_class {
    RT_K_RAYGEN = 0, RT_K_CLOSEST = 1, RT_K_SHADE = 2, RT_K_SHADOW = 3, RT_K_ACCUMULATE = 4, RT_K_RESOLVE = 5, RT_K_COUNT = 6
} rt_kernel_class;
typedef struct rt_kernel_profile {
    uint64_t launches[RT_K_COUNT];
    double   total_ms[RT_K_COUNT];
} rt_kernel_profile;
int rt_render_set_profiling(rt_render* render, int enable);          /* also clears the totals */
int rt_render_kernel_profile(rt_render* render, rt_kernel_profile* out);

/* ---- ABI 5 ---- The two warn-only validator rules per object (GraphicsValidator.Errors[type][object], GraphicsValidator.cs:18, 35-45;
 * recorded at RenderJob.cs:597 and :606): normals_wrong[i] / uv_invalid[i] = closest hits on scene object i whose normal is not a unit
 * vector / whose UV lies outside [0, 1]^2, since the render was created.  Needs RT_RENDER_VALIDATE (RT_ERR_INVALID_ARG without);
 * n_objects must be the scene's object count.  Either array may be NULL.  Summed over the shards when n_devices > 1. */
int rt_render_validator_counts(rt_render* render, uint64_t* normals_wrong, uint64_t* uv_invalid, uint32_t n_objects);

/* Parity hooks: RenderJob.TryFindClosestHit (RenderJob.cs:581-617) and AnyIntersectionFast (:575)
 * on caller-supplied rays. Ray i uses RNG key (seed, pixel=i, pass=0, depth=0) where an
 * intersector draws random numbers (ConstantDensityMedium.cs:56,63). */
int rt_trace_closest(rt_scene* scene, const rt_ray* rays, uint64_t n, float k_min, float k_max, uint64_t seed, rt_hit* hits);
int rt_trace_any(rt_scene* scene, const rt_ray* rays, const float* k_min, const float* k_max, uint64_t n, uint8_t* occluded);

/* rt_trace_closest for callers that cannot allocate device memory: tight serving loops, harnesses that forbid hipMalloc.  It has NO
 * counterpart in the reference.  Same arguments, same kernel and bit-identical results as rt_trace_closest, but the device copies of
 * `rays` and `hits` (rt_trace_closest keeps grow-only staging buffers in the scene: its first call, and every larger one, hipMallocs)
 * are staged in a fixed array inside the library's code object: (sizeof(rt_ray) + sizeof(rt_hit)) * RT_TRACE_NOALLOC_MAX_RAYS = 9.4 MB
 * per GPU, held from library load until the process ends.  The call performs no hipMalloc, hipFree or other allocating routine; its
 * two copies and its launch go to a non-blocking stream the library keeps per device (created with the scene), so it does not touch
 * the null stream.  n > RT_TRACE_NOALLOC_MAX_RAYS -> RT_ERR_TOO_LARGE (split the batch, or use rt_trace_closest).
 * Every call on one GPU shares that GPU's array (it keeps the previous call's contents; entries >= n are stale, never read):
 * calls that overlap on one GPU race — serialise them per device, whatever scenes they use. */
#define RT_TRACE_NOALLOC_MAX_RAYS (1u << 17)
int rt_trace_closest_noalloc(rt_scene* scene, const rt_ray* rays, uint64_t n, float k_min, float k_max, uint64_t seed, rt_hit* hits);

#ifdef __cplusplus
}
#endif
#endif /* RT_HIP_H */
