// HipRenderBackend.cs — the reference-side binding of include/rt_hip.h (source only: this image has no .NET SDK,
// so the file is not compiled here; tests/test_csharp_binding.py checks every [StructLayout] struct and every
// [DllImport] below against the header field by field, and the tested drivers of the same entry points are
// raytracer_amd/capi.py and tools/rtbench.cpp).
//
// Where it goes: Ararem.RayTracer.Core/Backends/HipRenderBackend.cs.  RenderJob.RenderInternal (RenderJob.cs:69-126)
// gets one new first line:
//
//     if (RenderOptions.Backend == RenderBackend.Hip) { new HipRenderBackend().Render(this, cancellationToken); return; }
//
// and RenderOptions three new init-only properties: `Backend` (default Cpu, or env RT_BACKEND=hip), `Devices`
// (HIP ordinals; default: one GPU), `ExactLightSampling`, `ExactOcclusion` and `Validate`.  Nothing else in the public surface changes: Image,
// ImageBuffer, RenderStats, RenderTask, StartOrGetRenderAsync keep their meaning.
//
// Core cannot name Impl types (Impl -> Core, Ararem.RayTracer.Impl.csproj:23), so flattening is an interface the
// builtin Impl classes implement (csharp/Impl/DeviceFlatteners.cs); anything that does not implement it is rejected
// with NotSupportedException at Render() time — there is no CPU fallback behind this backend.
using System;
using System.Collections.Concurrent;
using System.Collections.Generic;
using System.Runtime.CompilerServices;
using System.Runtime.InteropServices;
using System.Threading;
using Ararem.RayTracer.Core.Acceleration;
using Ararem.RayTracer.Core.Debugging;

namespace Ararem.RayTracer.Core.Backends;

public enum RenderBackend { Cpu = 0, Hip = 1 }

/// <summary>What RenderJob.RenderInternal hands the pass loop to (RenderJob.cs:69): the CPU loop stays the default implementation.</summary>
internal interface IRenderBackend
{
	/// <summary>Renders job.RenderOptions.Passes passes (or until cancelled) into job.Image / job.RenderStats.</summary>
	void Render(RenderJob job, CancellationToken ct);
}

/// <summary>Implemented by every builtin Hittable / Material / Texture / Light in Ararem.RayTracer.Impl.</summary>
public interface IDeviceFlattenable
{
	/// <summary>Appends this object's record(s) to the builder's tables and returns its index in its table.</summary>
	uint Flatten(DeviceSceneBuilder builder);
}

/// <summary>Implemented by the Hittables: their table (rt_hittable_type) besides the index, so that ConstantDensityMedium and
/// DiffuseShapedLight can refer to a shape that is not itself a scene object.</summary>
public interface IDeviceHittable : IDeviceFlattenable
{
	/// <summary>rt_hittable_type of the record Flatten appends.</summary>
	uint DeviceType { get; }
}

/// <summary>Implemented by DefaultSkyBox and SingleColourSkyBox.</summary>
public interface IDeviceSkyBox { RtHip.SkyBox FlattenSkyBox(); }

public static unsafe class RtHip
{
	private const string Lib = "rt_hip"; // librt_hip.so next to the assembly, or on LD_LIBRARY_PATH

	public const uint AbiVersion = 5;            // RT_ABI_VERSION
	public const uint NoHit      = 0xFFFFFFFFu;  // RT_NO_HIT
	public const int  MaxDevices = 16;           // RT_MAX_DEVICES
	public const uint RenderExactLightSampling = 1u;   // RT_RENDER_EXACT_LIGHT_SAMPLING
	public const uint RenderExactOcclusion = 2u;       // RT_RENDER_EXACT_OCCLUSION
	public const uint RenderValidate = 4u;             // RT_RENDER_VALIDATE
	public const uint TraceNoAllocMaxRays = 1u << 17;  // RT_TRACE_NOALLOC_MAX_RAYS

	// rt_hittable_type
	public const uint HitSphere = 0, HitQuad = 1, HitTriangle = 2, HitInfinitePlane = 3, HitXyPlane = 4, HitXzPlane = 5, HitYzPlane = 6, HitDisk = 7, HitBox = 8,
	                  HitCapsule = 9, HitCylinder = 10, HitMedium = 11;
	// rt_texture_type, rt_material_type, rt_light_type, rt_attenuation_kind, rt_skybox_type
	public const uint TexSolid = 0, TexMarble = 1, TexGreyscalePerlin = 2;
	public const uint MatStandard = 0, MatRefractive = 1, MatEmissiveRefractive = 2, MatPhong = 3, MatVolumetric = 4;
	public const uint MatFlagAlternateRefraction = 1u, MatFlagDirectEmission = 2u;
	public const uint LightPoint = 0, LightDiffuseSphere = 1, LightDiffuseShaped = 2;
	public const uint AttenConstant = 0, AttenLinear = 1, AttenPower = 2, AttenPowerStayHigh = 3, AttenLogistic = 4, AttenLogisticLkx = 5, AttenExpDecay = 6;
	public const uint SkyDefault = 0, SkySingleColour = 1;

	[StructLayout(LayoutKind.Sequential)] public struct Sphere  { public fixed float Centre[3]; public float Radius; }
	[StructLayout(LayoutKind.Sequential)] public struct Quad    { public fixed float Origin[3]; public fixed float Normal[3]; public fixed float InvCol1[4]; public fixed float InvCol2[4]; }
	[StructLayout(LayoutKind.Sequential)] public struct Triangle{ public fixed float V0[3]; public fixed float E1[3]; public fixed float E2[3]; public fixed float Normal[3]; public fixed float InvCol1[4]; public fixed float InvCol2[4]; }
	[StructLayout(LayoutKind.Sequential)] public struct InfinitePlane { public fixed float Point[3]; public fixed float Normal[3]; public float NegPointDotNormal; }
	[StructLayout(LayoutKind.Sequential)] public struct AxisPlane { public float ALo; public float AHi; public float BLo; public float BHi; public float C; }
	[StructLayout(LayoutKind.Sequential)] public struct Disk    { public fixed float Centre[3]; public fixed float Normal[3]; public float Radius; public float NegCentreDotNorm; public fixed float InvCol1[4]; public fixed float InvCol2[4]; }
	[StructLayout(LayoutKind.Sequential)] public struct Box     { public fixed float WorldToBox[12]; public fixed float BoxToWorldRows[9]; }
	[StructLayout(LayoutKind.Sequential)] public struct Capsule { public fixed float P1[3]; public fixed float P2[3]; public float Radius; public fixed float UvMatrix[9]; }
	[StructLayout(LayoutKind.Sequential)] public struct Cylinder{ public fixed float P1[3]; public fixed float P2[3]; public float Radius; }
	[StructLayout(LayoutKind.Sequential)] public struct Medium  { public uint BoundaryType; public uint BoundaryIndex; public float Density; public float NegInvDensity; }
	[StructLayout(LayoutKind.Sequential)] public struct Object  { public uint Type; public uint Index; public uint Material; public fixed float AabbMin[3]; public fixed float AabbMax[3]; }
	[StructLayout(LayoutKind.Sequential)] public struct Texture { public uint Type; public fixed float F[7]; }
	[StructLayout(LayoutKind.Sequential)] public struct Material{ public uint Type; public fixed uint Tex[2]; public uint Flags; public fixed float F[10]; }
	[StructLayout(LayoutKind.Sequential)] public struct Light
	{
		public uint Type; public fixed float Position[3]; public fixed float Colour[3];
		public float AttenuationRadius; public float CutoffRadius; public float DiffusionRadius;
		public uint AttenKind; public fixed float AttenP[3];
		public uint ShapeType; public uint ShapeIndex; public fixed float ShapeAabbMin[3]; public fixed float ShapeAabbMax[3];
	}
	[StructLayout(LayoutKind.Sequential)] public struct SkyBox  { public uint Type; public fixed float Colour[3]; }
	[StructLayout(LayoutKind.Sequential)] public struct Camera
	{
		public fixed float LookFrom[3]; public fixed float Horizontal[3]; public fixed float Vertical[3]; public fixed float LowerLeftCorner[3];
		public fixed float U[3]; public fixed float V[3]; public float LensRadius;
	}

	[StructLayout(LayoutKind.Sequential)]
	public struct SceneDesc
	{
		public uint StructSize; public uint AbiVersion;
		public Camera Camera; public SkyBox Skybox;
		public uint NObjects;        public Object*        Objects;
		public uint NSpheres;        public Sphere*        Spheres;
		public uint NQuads;          public Quad*          Quads;
		public uint NTriangles;      public Triangle*      Triangles;
		public uint NInfinitePlanes; public InfinitePlane* InfinitePlanes;
		public uint NXyPlanes;       public AxisPlane*     XyPlanes;
		public uint NXzPlanes;       public AxisPlane*     XzPlanes;
		public uint NYzPlanes;       public AxisPlane*     YzPlanes;
		public uint NDisks;          public Disk*          Disks;
		public uint NBoxes;          public Box*           Boxes;
		public uint NCapsules;       public Capsule*       Capsules;
		public uint NCylinders;      public Cylinder*      Cylinders;
		public uint NMedia;          public Medium*        Media;
		public uint NMaterials;      public Material*      Materials;
		public uint NTextures;       public Texture*       Textures;
		public uint NLights;         public Light*         Lights;
	}

	/// <summary>rt_scene_info: what BvhTree's construction log lines report (Acceleration/BvhTree.cs:29-41), plus the kernel variant and tree placement</summary>
	[StructLayout(LayoutKind.Sequential)]
	public struct SceneInfo
	{
		public uint NObjects; public uint NUnbounded; public uint NNodes; public uint BvhDepth; public uint TreeInLds; public uint LdsImageBytes; public uint BuiltOnDevice; public uint ObjectList;
		public fixed byte Variant[16];
		public uint WideNodes; public uint WideLevels; public int Device; public uint Reserved;
	}

	[StructLayout(LayoutKind.Sequential)]
	public struct RenderOpts
	{
		public uint StructSize; public uint Width; public uint Height;
		public float KMin; public float KMax;
		public uint MaxBounceDepth; public uint LightSampleCountHint; public uint DebugVisualisation;
		public ulong Seed;
		public uint PassesPerBatch; public int Device;
		public IntPtr Stream; public float* AccumDevice;
		public uint Flags; public uint MaxPassesPerCall; public uint NDevices;
		public fixed int DeviceIds[16];
	}

	[StructLayout(LayoutKind.Sequential)]
	public struct Stats
	{
		public ulong PixelsRendered; public ulong PassesRendered; public ulong MaterialScatterCount; public ulong MaterialAbsorbedCount; public ulong AabbMisses;
		public ulong HittableMisses; public ulong HittableIntersections; public ulong SkyRays; public ulong BounceLimitExceeded; public ulong RayCount;
		public ulong ShadowRays; public ulong NodeVisits; public ulong PrimTests; public ulong ShadowNodeVisits; public ulong ShadowPrimTests;
		public ulong NormalsWrongMagnitude; public ulong UvInvalid;
		public fixed ulong RayDepthCounts[64];
	}

	[StructLayout(LayoutKind.Sequential)] public struct Ray { public fixed float Origin[3]; public fixed float Direction[3]; }
	[StructLayout(LayoutKind.Sequential)]
	public struct Hit
	{
		public float K; public uint Object; public uint Material; public fixed float Normal[3]; public fixed float Uv[2]; public fixed float WorldPoint[3]; public uint OutsideFace;
	}
	[StructLayout(LayoutKind.Sequential)] public struct KernelProfile { public fixed ulong Launches[6]; public fixed double TotalMs[6]; }

	[DllImport(Lib)] public static extern int    rt_version();
	[DllImport(Lib)] public static extern IntPtr rt_last_error();
	[DllImport(Lib)] public static extern int    rt_device_count(out int n);
	[DllImport(Lib)] public static extern int    rt_scene_create(SceneDesc* desc, int device, out IntPtr scene);
	[DllImport(Lib)] public static extern int    rt_scene_destroy(IntPtr scene);
	[DllImport(Lib)] public static extern int    rt_scene_get_info(IntPtr scene, out SceneInfo info);
	[DllImport(Lib)] public static extern int    rt_render_create(IntPtr scene, RenderOpts* opts, out IntPtr render);
	[DllImport(Lib)] public static extern int    rt_render_destroy(IntPtr render);
	[DllImport(Lib)] public static extern int    rt_render_passes(IntPtr render, ulong firstPass, uint nPasses, uint passStride);
	[DllImport(Lib)] public static extern int    rt_render_read_rgb24(IntPtr render, byte* dst);
	[DllImport(Lib)] public static extern int    rt_render_read_accum(IntPtr render, float* rgbSum, out uint sampleCount);
	[DllImport(Lib)] public static extern int    rt_render_resolve_device(IntPtr render, float* accumDevice, uint sampleCount, byte* dstHost);
	[DllImport(Lib)] public static extern int    rt_render_stats(IntPtr render, out Stats stats);
	[DllImport(Lib)] public static extern int    rt_render_set_profiling(IntPtr render, int enable);
	[DllImport(Lib)] public static extern int    rt_render_kernel_profile(IntPtr render, out KernelProfile profile);
	[DllImport(Lib)] public static extern int    rt_render_validator_counts(IntPtr render, ulong* normalsWrong, ulong* uvInvalid, uint nObjects);
	[DllImport(Lib)] public static extern int    rt_trace_closest(IntPtr scene, Ray* rays, ulong n, float kMin, float kMax, ulong seed, Hit* hits);
	[DllImport(Lib)] public static extern int    rt_trace_any(IntPtr scene, Ray* rays, float* kMin, float* kMax, ulong n, byte* occluded);
	[DllImport(Lib)] public static extern int    rt_trace_closest_noalloc(IntPtr scene, Ray* rays, ulong n, float kMin, float kMax, ulong seed, Hit* hits);

	public static void Check(int rc)
	{
		if (rc == 0) return;
		string msg = Marshal.PtrToStringUTF8(rt_last_error()) ?? "unknown error";
		throw rc switch
		{
				-2 => new NotSupportedException(msg),     // RT_ERR_UNSUPPORTED: a type the kernels do not implement
				-1 => new ArgumentException(msg),         // RT_ERR_INVALID_ARG
				-5 => new OutOfMemoryException(msg),      // RT_ERR_OUT_OF_MEMORY
				-6 => new ArgumentOutOfRangeException(null, msg), // RT_ERR_TOO_LARGE
				_  => new InvalidOperationException($"rt_hip error {rc}: {msg}")
		};
	}

	public static void Put(float* dst, System.Numerics.Vector3 v) { dst[0] = v.X; dst[1] = v.Y; dst[2] = v.Z; }
	public static void Put(float* dst, Colour c) { dst[0] = c.R; dst[1] = c.G; dst[2] = c.B; }
}

/// <summary>Collects the flattened tables.  Ids are assigned by object reference: RefractiveMaterial.cs:51-52 and
/// EmissiveRefractiveMaterial.cs:53-63 compare materials with == / != on references, so two equal-valued materials stay distinct.</summary>
public sealed unsafe class DeviceSceneBuilder
{
	public readonly List<RtHip.Object> Objects = new();
	public readonly List<RtHip.Sphere> Spheres = new();
	public readonly List<RtHip.Quad> Quads = new();
	public readonly List<RtHip.Triangle> Triangles = new();
	public readonly List<RtHip.InfinitePlane> InfinitePlanes = new();
	public readonly List<RtHip.AxisPlane> XyPlanes = new(), XzPlanes = new(), YzPlanes = new();
	public readonly List<RtHip.Disk> Disks = new();
	public readonly List<RtHip.Box> Boxes = new();
	public readonly List<RtHip.Capsule> Capsules = new();
	public readonly List<RtHip.Cylinder> Cylinders = new();
	public readonly List<RtHip.Medium> Media = new();
	public readonly List<RtHip.Material> Materials = new();
	public readonly List<RtHip.Texture> Textures = new();
	public readonly List<RtHip.Light> Lights = new();
	private readonly Dictionary<object, uint> ids = new(ReferenceEqualityComparer.Instance);

	/// <summary>Index of a Material / Texture in its table, flattening it on first sight (one entry per distinct REFERENCE); rejects user-defined types.</summary>
	public uint IdOf(object o)
	{
		if (ids.TryGetValue(o, out uint id)) return id;
		if (o is not IDeviceFlattenable f)
			throw new NotSupportedException($"{o.GetType().FullName} has no device representation (user-defined scene types are rejected by the HIP backend, not emulated)");
		id = f.Flatten(this);
		ids[o] = id;
		return id;
	}

	/// <summary>(rt_hittable_type, index) of a Hittable's primitive record; a shape used twice (scene object and light shape) gets one record.</summary>
	public (uint type, uint index) Shape(Hittable h)
	{
		if (h is not IDeviceHittable d)
			throw new NotSupportedException($"{h.GetType().FullName} has no device representation (user-defined Hittables are rejected by the HIP backend, not emulated)");
		return (d.DeviceType, IdOf(h));
	}

	/// <summary>One Scene.SceneObjects element, in scene order: its index in Objects is the object id the kernels report.</summary>
	public void AddObject(Hittable hittable, Material material)
	{
		(uint type, uint index) = Shape(hittable);
		AxisAlignedBoundingBox aabb = hittable.BoundingVolume;
		RtHip.Object o = new() { Type = type, Index = index, Material = IdOf(material) };
		o.AabbMin[0] = aabb.Min.X; o.AabbMin[1] = aabb.Min.Y; o.AabbMin[2] = aabb.Min.Z;
		o.AabbMax[0] = aabb.Max.X; o.AabbMax[1] = aabb.Max.Y; o.AabbMax[2] = aabb.Max.Z;
		Objects.Add(o);
	}
}

internal sealed unsafe class HipRenderBackend : IRenderBackend
{
	/// <summary>How long one rt_render_passes call should take: the cancellation and Image / RenderStats refresh granularity.  The reference checks
	/// its token per pixel (RenderJob.cs:97,114-118) and the GUI reads the image at 30 Hz (RenderBufferPanel.cs:53-79); a GPU call cannot be
	/// interrupted, so the number of passes per call follows the measured time of the previous call instead of being a constant (8 passes
	/// are 5 ms of Night Spheres at 1080p and 2 s of 100 000 spheres under 64 lights at 4K).</summary>
	private const double TargetSecondsPerCall = 0.1;
	/// <summary>Upper bound of passes per call and device, given to the library as max_passes_per_call: it keeps path state for this many passes per
	/// batch (184 B per path and batch slot), and a call of two batches keeps both of its batch slots busy.</summary>
	private const uint MaxPassesPerCall = 64;

	public void Render(RenderJob job, CancellationToken ct)
	{
		Scene scene = job.Scene; RenderOptions o = job.RenderOptions;
		DeviceSceneBuilder b = new();
		foreach (SceneObject so in scene.SceneObjects)
		{
			// the Hittable's own Material (SingleMaterialHittable.Material, ConstantDensityMedium's internal VolumetricMaterial)
			if (so.Hittable is not IDeviceHittable) throw new NotSupportedException($"{so.Hittable.GetType().FullName}: user-defined Hittables are rejected by the HIP backend");
			b.AddObject(so.Hittable, DeviceMaterialOf(so.Hittable));
		}
		foreach (Light l in scene.Lights) b.IdOf(l);

		int[] devices = o.Devices is { Length: > 0 } ? o.Devices : new[] { -1 };
		if (devices.Length > RtHip.MaxDevices) throw new ArgumentOutOfRangeException(nameof(o.Devices), $"at most {RtHip.MaxDevices} devices");

		IntPtr hScene = IntPtr.Zero, hRender = IntPtr.Zero;
		RtHip.Object[] objs = b.Objects.ToArray(); RtHip.Sphere[] sph = b.Spheres.ToArray(); RtHip.Quad[] quads = b.Quads.ToArray(); RtHip.Triangle[] tris = b.Triangles.ToArray();
		RtHip.InfinitePlane[] planes = b.InfinitePlanes.ToArray(); RtHip.AxisPlane[] xy = b.XyPlanes.ToArray(), xz = b.XzPlanes.ToArray(), yz = b.YzPlanes.ToArray();
		RtHip.Disk[] disks = b.Disks.ToArray(); RtHip.Box[] boxes = b.Boxes.ToArray(); RtHip.Capsule[] caps = b.Capsules.ToArray(); RtHip.Cylinder[] cyls = b.Cylinders.ToArray();
		RtHip.Medium[] media = b.Media.ToArray(); RtHip.Material[] mats = b.Materials.ToArray(); RtHip.Texture[] texs = b.Textures.ToArray(); RtHip.Light[] lights = b.Lights.ToArray();
		try
		{
			fixed (RtHip.Object* pO = objs) fixed (RtHip.Sphere* pS = sph) fixed (RtHip.Quad* pQ = quads) fixed (RtHip.Triangle* pT = tris) fixed (RtHip.InfinitePlane* pP = planes)
			fixed (RtHip.AxisPlane* pXy = xy, pXz = xz, pYz = yz) fixed (RtHip.Disk* pD = disks) fixed (RtHip.Box* pB = boxes) fixed (RtHip.Capsule* pC = caps) fixed (RtHip.Cylinder* pCy = cyls)
			fixed (RtHip.Medium* pM = media) fixed (RtHip.Material* pMat = mats) fixed (RtHip.Texture* pTex = texs) fixed (RtHip.Light* pL = lights)
			{
				RtHip.SceneDesc d = new()
				{
						StructSize = (uint)sizeof(RtHip.SceneDesc), AbiVersion = RtHip.AbiVersion,
						Camera = FlattenCamera(scene.Camera), Skybox = FlattenSky(scene.SkyBox),
						NObjects = (uint)objs.Length, Objects = pO, NSpheres = (uint)sph.Length, Spheres = pS, NQuads = (uint)quads.Length, Quads = pQ,
						NTriangles = (uint)tris.Length, Triangles = pT, NInfinitePlanes = (uint)planes.Length, InfinitePlanes = pP,
						NXyPlanes = (uint)xy.Length, XyPlanes = pXy, NXzPlanes = (uint)xz.Length, XzPlanes = pXz, NYzPlanes = (uint)yz.Length, YzPlanes = pYz,
						NDisks = (uint)disks.Length, Disks = pD, NBoxes = (uint)boxes.Length, Boxes = pB, NCapsules = (uint)caps.Length, Capsules = pC,
						NCylinders = (uint)cyls.Length, Cylinders = pCy, NMedia = (uint)media.Length, Media = pM,
						NMaterials = (uint)mats.Length, Materials = pMat, NTextures = (uint)texs.Length, Textures = pTex, NLights = (uint)lights.Length, Lights = pL
				};
				RtHip.Check(RtHip.rt_scene_create(&d, devices[0], out hScene));       // the library copies every table during the call
			}
			RtHip.RenderOpts ro = new()
			{
					StructSize = (uint)sizeof(RtHip.RenderOpts), Width = (uint)o.RenderWidth, Height = (uint)o.RenderHeight, KMin = o.KMin, KMax = o.KMax,
					MaxBounceDepth = (uint)o.MaxBounceDepth, LightSampleCountHint = (uint)o.LightSampleCountHint, DebugVisualisation = (uint)o.DebugVisualisation,   // rt_debug_vis has the enum's values
					Seed = (ulong)Random.Shared.NextInt64(), PassesPerBatch = 0, Device = -1, Stream = IntPtr.Zero, AccumDevice = null,
					Flags = (o.ExactLightSampling ? RtHip.RenderExactLightSampling : 0u) | (o.ExactOcclusion ? RtHip.RenderExactOcclusion : 0u) | (o.Validate ? RtHip.RenderValidate : 0u),
					MaxPassesPerCall = MaxPassesPerCall * (uint)devices.Length,     // one call's passes are split over the devices
					NDevices = devices.Length > 1 ? (uint)devices.Length : 0u
			};
			if (devices.Length > 1)
			{
				RtHip.SceneInfo _; RtHip.Check(RtHip.rt_scene_get_info(hScene, out _));
				for (int i = 0; i < devices.Length; i++) ro.DeviceIds[i] = devices[i];   // device_ids[0] must be the scene's device: devices[0] above
			}
			RtHip.Check(RtHip.rt_render_create(hScene, &ro, out hRender));

			uint nDev = (uint)devices.Length, perCall = nDev;                    // the first call: one pass per device, timed
			byte[] rgb = new byte[job.ImageBuffer.Width * job.ImageBuffer.Height * 3];   // one staging buffer for every Publish of this job
			ulong[] normalsWrong = o.Validate ? new ulong[objs.Length] : Array.Empty<ulong>(), uvInvalid = o.Validate ? new ulong[objs.Length] : Array.Empty<ulong>();
			System.Diagnostics.Stopwatch call = new();
			job.Stopwatch.Start();
			for (ulong pass = 0; o.InfinitePasses || pass < o.Passes;)
			{
				if (ct.IsCancellationRequested) break;                           // RenderJob.cs:114-118: log and leave the loop, the task completes normally
				uint n = o.InfinitePasses ? perCall : (uint)Math.Min((ulong)perCall, o.Passes - pass);
				call.Restart();
				RtHip.Check(RtHip.rt_render_passes(hRender, pass, n, 1));          // with n_devices > 1 the library shards these n passes over the GPUs
				double seconds = Math.Max(call.Elapsed.TotalSeconds, 1e-6);
				pass += n;
				Publish(job, hRender, rgb, normalsWrong, uvInvalid);               // the GUIs poll Image / RenderStats at 30 Hz without locks (RenderBufferPanel.cs:53-79)
				// next call: as many passes as fit TargetSecondsPerCall at the rate just seen, a multiple of the device count, at most the library's batch
				// (per device a multiple of 8 once there is room for 8, of 4 or 2 below: the library traces a batch's passes in groups of the largest
				//  power of two that divides their number, DESIGN.md §4)
				double want = n * TargetSecondsPerCall / seconds;
				uint each = (uint)Math.Clamp(Math.Floor(want / nDev), 1.0, (double)MaxPassesPerCall);
				each = each >= 8 ? each & ~7u : each >= 4 ? 4u : each >= 2 ? 2u : 1u;
				perCall = each * nDev;
			}
			job.Stopwatch.Stop();
		}
		finally
		{
			if (hRender != IntPtr.Zero) RtHip.rt_render_destroy(hRender);          // renders before their scene: rt_scene_destroy refuses otherwise
			if (hScene != IntPtr.Zero) RtHip.rt_scene_destroy(hScene);
		}
	}

	/// <summary>The Material a Hittable's HitRecords carry: SingleMaterialHittable.Material (SingleMaterialHittable.cs:11) or the medium's private
	/// VolumetricMaterial (ConstantDensityMedium.cs:17), which ConstantDensityMedium exposes to the backend through IDeviceMaterialOwner.</summary>
	private static Material DeviceMaterialOf(Hittable h) => h switch
	{
			IDeviceMaterialOwner m => m.DeviceMaterial,
			_                      => throw new NotSupportedException($"{h.GetType().FullName} does not expose its Material to the HIP backend")
	};

	/// <summary>RGB24 rows (row 0 = top, as Image&lt;Rgb24&gt;) into ImageBuffer; counters into RenderStats (RenderStats.cs:12-93); with
	/// RenderOptions.Validate the two warn-only validator rules into GraphicsValidator.Errors per SceneObject, as RenderJob.cs:597,606 files them.</summary>
	private static void Publish(RenderJob job, IntPtr hRender, byte[] rgb, ulong[] normalsWrong, ulong[] uvInvalid)
	{
		int w = job.ImageBuffer.Width, h = job.ImageBuffer.Height;
		fixed (byte* p = rgb) RtHip.Check(RtHip.rt_render_read_rgb24(hRender, p));
		job.ImageBuffer.ProcessPixelRows(acc =>
		{
			for (int y = 0; y < h; y++) MemoryMarshal.Cast<byte, SixLabors.ImageSharp.PixelFormats.Rgb24>(rgb.AsSpan(y * w * 3, w * 3)).CopyTo(acc.GetRowSpan(y));
		});
		RtHip.Check(RtHip.rt_render_stats(hRender, out RtHip.Stats s));
		RenderStats rs = job.RenderStats;
		rs.PixelsRendered = s.PixelsRendered; rs.PassesRendered = s.PassesRendered; rs.MaterialScatterCount = s.MaterialScatterCount;
		rs.MaterialAbsorbedCount = s.MaterialAbsorbedCount; rs.SkyRays = s.SkyRays; rs.BounceLimitExceeded = s.BounceLimitExceeded; rs.RayCount = s.RayCount;
		rs.AabbMisses = s.AabbMisses; rs.HittableMisses = s.HittableMisses; rs.HittableIntersections = s.HittableIntersections;   // RenderStats.cs: the three BVH counters the GUI panels show
		for (int d = 0; d < 64; d++) if (s.RayDepthCounts[d] != 0) rs.RayDepthCounts[(ulong)d] = s.RayDepthCounts[d];          // ConcurrentDictionary<ulong, ulong> (RenderStats.cs:93)
		ConcurrentDictionary<object, ulong> Filed(GraphicsErrorType t) => GraphicsValidator.Errors.GetOrAdd(t, _ => new ConcurrentDictionary<object, ulong>());
		if (normalsWrong.Length != 0)
		{
			// per object, keyed by the scene object as the reference keys them (RenderJob.cs:597, :606 pass `obj`; scene order = the library's object ids)
			fixed (ulong* pn = normalsWrong) fixed (ulong* pu = uvInvalid) RtHip.Check(RtHip.rt_render_validator_counts(hRender, pn, pu, (uint)normalsWrong.Length));
			for (int i = 0; i < normalsWrong.Length; i++)
			{
				if (normalsWrong[i] != 0) Filed(GraphicsErrorType.NormalsWrongMagnitude)[job.Scene.SceneObjects[i]] = normalsWrong[i];
				if (uvInvalid[i] != 0) Filed(GraphicsErrorType.UVInvalid)[job.Scene.SceneObjects[i]] = uvInvalid[i];
			}
		}
		else if (s.NormalsWrongMagnitude != 0) Filed(GraphicsErrorType.NormalsWrongMagnitude)[job.Scene] = s.NormalsWrongMagnitude;   // without Validate the device keeps the scene's total only
	}

	private static RtHip.Camera FlattenCamera(Camera c)
	{
		RtHip.Camera r = new() { LensRadius = c.LensRadius };
		RtHip.Put(r.LookFrom, c.LookFrom); RtHip.Put(r.Horizontal, c.Horizontal); RtHip.Put(r.Vertical, c.Vertical); RtHip.Put(r.LowerLeftCorner, c.LowerLeftCorner);
		RtHip.Put(r.U, c.U); RtHip.Put(r.V, c.V);
		return r;
	}

	private static RtHip.SkyBox FlattenSky(SkyBox sky)
	{
		if (sky is not IDeviceSkyBox s) throw new NotSupportedException($"{sky.GetType().FullName}: user-defined SkyBoxes are rejected by the HIP backend");
		return s.FlattenSkyBox();
	}
}

/// <summary>Implemented by SingleMaterialHittable (returns Material) and ConstantDensityMedium (returns its internal VolumetricMaterial).</summary>
public interface IDeviceMaterialOwner { Material DeviceMaterial { get; } }
