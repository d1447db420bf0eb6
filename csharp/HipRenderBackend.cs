// HipRenderBackend.cs — the reference-side binding of include/rt_hip.h (source only: this image has no .NET SDK,
// so the file is not compiled here; the tested drivers of the same entry points are raytracer_amd/capi.py and tests/).
//
// Where it goes: Ararem.RayTracer.Core/Backends/HipRenderBackend.cs.  RenderJob.RenderInternal (RenderJob.cs:69-126)
// gets one new first line:
//
//     if (RenderOptions.Backend == RenderBackend.Hip) { new HipRenderBackend().Render(this, cancellationToken); return; }
//
// and RenderOptions one new init-only property `Backend` (default Cpu, or env RT_BACKEND=hip).  Nothing else in the
// public surface changes: Image, ImageBuffer, RenderStats, RenderTask, StartOrGetRenderAsync keep their meaning.
//
// Core cannot name Impl types (Impl -> Core, Ararem.RayTracer.Impl.csproj:23), so flattening is an interface the
// builtin Impl classes implement; anything that does not implement it is rejected with NotSupportedException at
// Render() time — there is no CPU fallback behind this backend.
using System;
using System.Collections.Generic;
using System.Runtime.CompilerServices;
using System.Runtime.InteropServices;
using System.Threading;
using Ararem.RayTracer.Core.Acceleration;

namespace Ararem.RayTracer.Core.Backends;

public enum RenderBackend { Cpu = 0, Hip = 1 }

/// <summary>Implemented by every builtin Hittable / Material / Texture / Light / SkyBox in Ararem.RayTracer.Impl.</summary>
public interface IDeviceFlattenable
{
	/// <summary>Appends this object's record(s) to the builder's tables and returns its index in its table.</summary>
	uint Flatten(DeviceSceneBuilder builder);
}

internal static unsafe class RtHip
{
	private const string Lib = "rt_hip"; // librt_hip.so next to the assembly, or on LD_LIBRARY_PATH

	public const uint AbiVersion = 1;
	public const uint NoHit      = 0xFFFFFFFFu;

	[StructLayout(LayoutKind.Sequential)] public struct Object  { public uint Type, Index, Material; public fixed float AabbMin[3]; public fixed float AabbMax[3]; }
	[StructLayout(LayoutKind.Sequential)] public struct Sphere  { public fixed float Centre[3]; public float Radius; }
	[StructLayout(LayoutKind.Sequential)] public struct Quad    { public fixed float Origin[3]; public fixed float Normal[3]; public fixed float InvCol1[4]; public fixed float InvCol2[4]; }
	[StructLayout(LayoutKind.Sequential)] public struct Triangle{ public fixed float V0[3]; public fixed float E1[3]; public fixed float E2[3]; public fixed float Normal[3]; public fixed float InvCol1[4]; public fixed float InvCol2[4]; }
	[StructLayout(LayoutKind.Sequential)] public struct InfinitePlane { public fixed float Point[3]; public fixed float Normal[3]; public float NegPointDotNormal; }
	[StructLayout(LayoutKind.Sequential)] public struct AxisPlane { public float ALo, AHi, BLo, BHi, C; }
	[StructLayout(LayoutKind.Sequential)] public struct Disk    { public fixed float Centre[3]; public fixed float Normal[3]; public float Radius, NegCentreDotNorm; public fixed float InvCol1[4]; public fixed float InvCol2[4]; }
	[StructLayout(LayoutKind.Sequential)] public struct Box     { public fixed float WorldToBox[12]; public fixed float BoxToWorldRows[9]; }
	[StructLayout(LayoutKind.Sequential)] public struct Capsule { public fixed float P1[3]; public fixed float P2[3]; public float Radius; public fixed float UvMatrix[9]; }
	[StructLayout(LayoutKind.Sequential)] public struct Cylinder{ public fixed float P1[3]; public fixed float P2[3]; public float Radius; }
	[StructLayout(LayoutKind.Sequential)] public struct Medium  { public uint BoundaryType, BoundaryIndex; public float Density, NegInvDensity; }
	[StructLayout(LayoutKind.Sequential)] public struct Texture { public uint Type; public fixed float F[7]; }
	[StructLayout(LayoutKind.Sequential)] public struct Material{ public uint Type; public fixed uint Tex[2]; public uint Flags; public fixed float F[10]; }
	[StructLayout(LayoutKind.Sequential)] public struct Light
	{
		public uint Type; public fixed float Position[3]; public fixed float Colour[3];
		public float AttenuationRadius, CutoffRadius, DiffusionRadius;
		public uint AttenKind; public fixed float AttenP[3];
		public uint ShapeType, ShapeIndex; public fixed float ShapeAabbMin[3]; public fixed float ShapeAabbMax[3];
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
		public uint StructSize, AbiVersion;
		public Camera Camera; public SkyBox SkyBox;
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
	public unsafe struct SceneInfo
	{
		public uint NObjects, NUnbounded, NNodes, BvhDepth, TreeInLds, LdsImageBytes, BuiltOnDevice, ObjectList;
		public fixed byte Variant[16];
	}

	[StructLayout(LayoutKind.Sequential)]
	public struct RenderOpts
	{
		public uint StructSize, Width, Height;
		public float KMin, KMax;
		public uint MaxBounceDepth, LightSampleCountHint, DebugVisualisation;
		public ulong Seed;
		public uint PassesPerBatch; public int Device;
		public IntPtr Stream; public float* AccumDevice;
	}

	[StructLayout(LayoutKind.Sequential)]
	public struct Stats
	{
		public ulong PixelsRendered, PassesRendered, MaterialScatterCount, MaterialAbsorbedCount, AabbMisses, HittableMisses, HittableIntersections,
		             SkyRays, BounceLimitExceeded, RayCount, ShadowRays, NodeVisits, PrimTests, ShadowNodeVisits, ShadowPrimTests;
		public fixed ulong RayDepthCounts[64];
	}

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
	[DllImport(Lib)] public static extern int    rt_render_stats(IntPtr render, out Stats stats);

	public static void Check(int rc)
	{
		if (rc == 0) return;
		string msg = Marshal.PtrToStringUTF8(rt_last_error()) ?? "unknown error";
		throw rc switch
		{
				-2 => new NotSupportedException(msg),     // RT_ERR_UNSUPPORTED: a type the kernels do not implement
				-1 => new ArgumentException(msg),         // RT_ERR_INVALID_ARG
				-5 => new OutOfMemoryException(msg),
				_  => new InvalidOperationException($"rt_hip error {rc}: {msg}")
		};
	}
}

/// <summary>Collects the flattened tables.  Ids are assigned by object reference: RefractiveMaterial.cs:51-52 and
/// EmissiveRefractiveMaterial.cs:53-63 compare materials with == / != on references, so two equal-valued materials stay distinct.</summary>
public sealed unsafe class DeviceSceneBuilder
{
	internal readonly List<RtHip.Object> Objects = new();
	internal readonly List<RtHip.Sphere> Spheres = new();
	internal readonly List<RtHip.Quad> Quads = new();
	internal readonly List<RtHip.Triangle> Triangles = new();
	internal readonly List<RtHip.InfinitePlane> InfinitePlanes = new();
	internal readonly List<RtHip.AxisPlane> XyPlanes = new(), XzPlanes = new(), YzPlanes = new();
	internal readonly List<RtHip.Disk> Disks = new();
	internal readonly List<RtHip.Box> Boxes = new();
	internal readonly List<RtHip.Capsule> Capsules = new();
	internal readonly List<RtHip.Cylinder> Cylinders = new();
	internal readonly List<RtHip.Medium> Media = new();
	internal readonly List<RtHip.Material> Materials = new();
	internal readonly List<RtHip.Texture> Textures = new();
	internal readonly List<RtHip.Light> Lights = new();
	private readonly Dictionary<object, uint> ids = new(ReferenceEqualityComparer.Instance);

	/// <summary>Index of <paramref name="o"/> in its table, flattening it on first sight; rejects user-defined types.</summary>
	public uint IdOf(object o)
	{
		if (ids.TryGetValue(o, out uint id)) return id;
		if (o is not IDeviceFlattenable f)
			throw new NotSupportedException($"{o.GetType().FullName} has no device representation (user-defined scene types are rejected by the HIP backend, not emulated)");
		id = f.Flatten(this);
		ids[o] = id;
		return id;
	}

	/// <summary>One Scene.SceneObjects element: <paramref name="type"/>/<paramref name="index"/> name the record its Hittable appended.</summary>
	public void AddObject(uint type, uint index, Material material, AxisAlignedBoundingBox aabb)
	{
		RtHip.Object o = new() { Type = type, Index = index, Material = IdOf(material) };
		o.AabbMin[0] = aabb.Min.X; o.AabbMin[1] = aabb.Min.Y; o.AabbMin[2] = aabb.Min.Z;
		o.AabbMax[0] = aabb.Max.X; o.AabbMax[1] = aabb.Max.Y; o.AabbMax[2] = aabb.Max.Z;
		Objects.Add(o);
	}
}

internal sealed unsafe class HipRenderBackend
{
	/// <summary>Passes traced per rt_render_passes call: the cancellation and Image/RenderStats refresh granularity.</summary>
	private const uint PassesPerCall = 8;

	public void Render(RenderJob job, CancellationToken ct)
	{
		Scene scene = job.Scene; RenderOptions o = job.RenderOptions;
		DeviceSceneBuilder b = new();
		foreach (SceneObject so in scene.SceneObjects) b.IdOf(so.Hittable);   // each Hittable.Flatten calls b.AddObject(...)
		foreach (Light l in scene.Lights) b.IdOf(l);

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
						Camera = FlattenCamera(scene.Camera), SkyBox = FlattenSky(scene.SkyBox),
						NObjects = (uint)objs.Length, Objects = pO, NSpheres = (uint)sph.Length, Spheres = pS, NQuads = (uint)quads.Length, Quads = pQ,
						NTriangles = (uint)tris.Length, Triangles = pT, NInfinitePlanes = (uint)planes.Length, InfinitePlanes = pP,
						NXyPlanes = (uint)xy.Length, XyPlanes = pXy, NXzPlanes = (uint)xz.Length, XzPlanes = pXz, NYzPlanes = (uint)yz.Length, YzPlanes = pYz,
						NDisks = (uint)disks.Length, Disks = pD, NBoxes = (uint)boxes.Length, Boxes = pB, NCapsules = (uint)caps.Length, Capsules = pC,
						NCylinders = (uint)cyls.Length, Cylinders = pCy, NMedia = (uint)media.Length, Media = pM,
						NMaterials = (uint)mats.Length, Materials = pMat, NTextures = (uint)texs.Length, Textures = pTex, NLights = (uint)lights.Length, Lights = pL
				};
				RtHip.Check(RtHip.rt_scene_create(&d, -1, out hScene));       // the library copies every table during the call
			}
			RtHip.RenderOpts ro = new()
			{
					StructSize = (uint)sizeof(RtHip.RenderOpts), Width = (uint)o.RenderWidth, Height = (uint)o.RenderHeight, KMin = o.KMin, KMax = o.KMax,
					MaxBounceDepth = (uint)o.MaxBounceDepth, LightSampleCountHint = (uint)o.LightSampleCountHint, DebugVisualisation = (uint)o.DebugVisualisation,   // rt_debug_vis has the enum's values
					Seed = (ulong)Random.Shared.NextInt64(), PassesPerBatch = 0, Device = -1, Stream = IntPtr.Zero, AccumDevice = null
			};
			RtHip.Check(RtHip.rt_render_create(hScene, &ro, out hRender));

			job.Stopwatch.Start();
			for (ulong pass = 0; o.InfinitePasses || pass < o.Passes;)
			{
				if (ct.IsCancellationRequested) break;                           // RenderJob.cs:114-118: log and leave the loop, the task completes normally
				uint n = o.InfinitePasses ? PassesPerCall : (uint)Math.Min(PassesPerCall, o.Passes - pass);
				RtHip.Check(RtHip.rt_render_passes(hRender, pass, n, 1));
				pass += n;
				Publish(job, hRender);                                             // the GUIs poll Image / RenderStats at 30 Hz without locks (RenderBufferPanel.cs:53-79)
			}
			job.Stopwatch.Stop();
		}
		finally
		{
			if (hRender != IntPtr.Zero) RtHip.rt_render_destroy(hRender);
			if (hScene != IntPtr.Zero) RtHip.rt_scene_destroy(hScene);
		}
	}

	/// <summary>RGB24 rows (row 0 = top, as Image&lt;Rgb24&gt;) into ImageBuffer; counters into RenderStats (RenderStats.cs:12-93).</summary>
	private static void Publish(RenderJob job, IntPtr hRender)
	{
		int w = job.ImageBuffer.Width, h = job.ImageBuffer.Height;
		byte[] rgb = new byte[w * h * 3];
		fixed (byte* p = rgb) RtHip.Check(RtHip.rt_render_read_rgb24(hRender, p));
		job.ImageBuffer.ProcessPixelRows(acc =>
		{
			for (int y = 0; y < h; y++) MemoryMarshal.Cast<byte, SixLabors.ImageSharp.PixelFormats.Rgb24>(rgb.AsSpan(y * w * 3, w * 3)).CopyTo(acc.GetRowSpan(y));
		});
		RtHip.Check(RtHip.rt_render_stats(hRender, out RtHip.Stats s));
		RenderStats rs = job.RenderStats;
		rs.PixelsRendered = s.PixelsRendered; rs.PassesRendered = s.PassesRendered; rs.MaterialScatterCount = s.MaterialScatterCount;
		rs.MaterialAbsorbedCount = s.MaterialAbsorbedCount; rs.SkyRays = s.SkyRays; rs.BounceLimitExceeded = s.BounceLimitExceeded; rs.RayCount = s.RayCount;
		for (int d = 0; d < 64; d++) if (s.RayDepthCounts[d] != 0) rs.RayDepthCounts[d] = s.RayDepthCounts[d];
	}

	private static RtHip.Camera FlattenCamera(Camera c)
	{
		RtHip.Camera r = new() { LensRadius = c.LensRadius };
		Put(r.LookFrom, c.LookFrom); Put(r.Horizontal, c.Horizontal); Put(r.Vertical, c.Vertical); Put(r.LowerLeftCorner, c.LowerLeftCorner); Put(r.U, c.U); Put(r.V, c.V);
		return r;
	}

	private static RtHip.SkyBox FlattenSky(SkyBox sky)
	{
		// DefaultSkyBox -> type 0, SingleColourSkyBox -> type 1 + colour; both implement IDeviceFlattenable in Impl and fill a RtHip.SkyBox
		if (sky is not IDeviceSkyBox s) throw new NotSupportedException($"{sky.GetType().FullName}: user-defined SkyBoxes are rejected by the HIP backend");
		return s.FlattenSkyBox();
	}

	[MethodImpl(MethodImplOptions.AggressiveInlining)]
	private static void Put(float* dst, System.Numerics.Vector3 v) { dst[0] = v.X; dst[1] = v.Y; dst[2] = v.Z; }
}

/// <summary>Implemented by DefaultSkyBox and SingleColourSkyBox.</summary>
public interface IDeviceSkyBox { internal RtHip.SkyBox FlattenSkyBox(); }

// Example of the per-type flatteners added to Ararem.RayTracer.Impl (one per builtin class; Sphere.cs:17-22 shown):
//
//   public sealed partial class Sphere : IDeviceFlattenable
//   {
//       uint IDeviceFlattenable.Flatten(DeviceSceneBuilder b)
//       {
//           RtHip.Sphere s = new() { Radius = Radius }; s.Centre[0] = Centre.X; s.Centre[1] = Centre.Y; s.Centre[2] = Centre.Z;
//           b.Spheres.Add(s);
//           uint index = (uint)b.Spheres.Count - 1;
//           b.AddObject(0 /* RT_HIT_SPHERE */, index, Material, BoundingVolume);
//           return index;
//       }
//   }
//
// SimpleLightBase.DistanceAttenuationFunc is a delegate (SimpleLightBase.cs:132-139): the factories at :83-129 are changed to
// return a small record { Kind, P0, P1, P2, Func } so the flattener reads Kind/P* instead of probing the lambda; a light built
// with a hand-written lambda has Kind = Custom and is rejected.
