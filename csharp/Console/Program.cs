// Program.cs — headless console front end (SURVEY §8f item 2: the role the reference's missing "SpectreConsole" project has beside
// its GUIs).  Source only: this image has no .NET SDK; tests/test_csharp_binding.py checks it against what it names.
//
// Where it goes: a new project Ararem.RayTracer.Display.Console referencing Core and Impl.
//
//   dotnet run -- --list
//   dotnet run -- --scene "Demo" --width 1920 --height 1080 --passes 256 --depth 10 --backend hip --devices 0,1 --out image.png
//
// Scene pick = BuiltinScenes.GetAll() by name (BuiltinScenes.cs:298-301); options = RenderOptions' init properties
// (Core/RenderOptions.cs:41-140) plus the three this backend adds (Backend, Devices, ExactLightSampling / ExactOcclusion:
// csharp/HipRenderBackend.cs); the job runs as every front end runs it (RenderJob.StartOrGetRenderAsync, RenderJob.cs:708), progress is
// polled from RenderStats the way the GUIs poll it (RenderStatsPanel reads the fields without locks), Ctrl-C cancels through the
// job's token (the backend stops after the batch in flight and the image so far is saved), and the picture is written as
// MainForm does (Display.EtoForms/MainForm.cs:124-129).  The Python twin of this file is raytracer_amd/__main__.py.
using Ararem.RayTracer.Core;
using Ararem.RayTracer.Core.Backends;
using Ararem.RayTracer.Core.Debugging;
using Ararem.RayTracer.Impl.Builtin;
using SixLabors.ImageSharp;
using SixLabors.ImageSharp.Formats.Png;
using System.Diagnostics;
using System.Globalization;

namespace Ararem.RayTracer.Display.Console;

internal static class Program
{
	private static int Main(string[] args)
	{
		Dictionary<string, string> opt = new(StringComparer.OrdinalIgnoreCase);
		HashSet<string> flags = new(StringComparer.OrdinalIgnoreCase);
		for (int i = 0; i < args.Length; i++)
		{
			if (!args[i].StartsWith("--")) return Usage($"unexpected argument '{args[i]}'");
			string key = args[i][2..];
			if (key is "list" or "exact-lights" or "exact-occlusion" or "validate" or "help") flags.Add(key);
			else if (i + 1 < args.Length) opt[key] = args[++i];
			else return Usage($"--{key} needs a value");
		}
		if (flags.Contains("help")) return Usage(null);

		Scene[] scenes = BuiltinScenes.GetAll().ToArray();
		if (flags.Contains("list"))
		{
			foreach (Scene s in scenes) System.Console.WriteLine(s);          // Scene.ToString(): "Name - N objects" (Scene.cs:12)
			return 0;
		}
		string wanted = opt.GetValueOrDefault("scene", "Demo");
		Scene? scene  = scenes.FirstOrDefault(s => string.Equals(s.Name, wanted, StringComparison.OrdinalIgnoreCase));
		if (scene is null) return Usage($"no builtin scene is called '{wanted}' (--list shows them)");

		RenderOptions options;
		try
		{
			options = new RenderOptions
			{
					RenderWidth          = ulong.Parse(opt.GetValueOrDefault("width", "1920"), CultureInfo.InvariantCulture),
					RenderHeight         = ulong.Parse(opt.GetValueOrDefault("height", "1080"), CultureInfo.InvariantCulture),
					Passes               = ulong.Parse(opt.GetValueOrDefault("passes", "100"), CultureInfo.InvariantCulture),
					MaxBounceDepth       = ulong.Parse(opt.GetValueOrDefault("depth", "10"), CultureInfo.InvariantCulture),
					LightSampleCountHint = ulong.Parse(opt.GetValueOrDefault("light-samples", "2"), CultureInfo.InvariantCulture),
					KMin                 = float.Parse(opt.GetValueOrDefault("kmin", "0.001"), CultureInfo.InvariantCulture),
					KMax                 = float.Parse(opt.GetValueOrDefault("kmax", "Infinity"), CultureInfo.InvariantCulture),
					DebugVisualisation   = Enum.Parse<GraphicsDebugVisualisation>(opt.GetValueOrDefault("vis", nameof(GraphicsDebugVisualisation.None)), true),
					Backend              = Enum.Parse<RenderBackend>(opt.GetValueOrDefault("backend", Environment.GetEnvironmentVariable("RT_BACKEND") ?? "hip"), true),
					Devices              = opt.GetValueOrDefault("devices", "0").Split(',', StringSplitOptions.RemoveEmptyEntries).Select(d => int.Parse(d, CultureInfo.InvariantCulture)).ToArray(),
					ExactLightSampling   = flags.Contains("exact-lights"),
					ExactOcclusion       = flags.Contains("exact-occlusion"),
					Validate             = flags.Contains("validate")
			};
		}
		catch (Exception e) when (e is FormatException or OverflowException or ArgumentException)      // ArgumentOutOfRangeException<T> from the init accessors included
		{
			return Usage(e.Message);
		}

		using CancellationTokenSource cancel = new();
		System.Console.CancelKeyPress += (_, ev) => { ev.Cancel = true; cancel.Cancel(); };
		RenderJob job   = new(scene, options);
		Stopwatch clock = Stopwatch.StartNew();
		Task      task  = job.StartOrGetRenderAsync(cancel.Token);
		while (!task.Wait(TimeSpan.FromSeconds(1)))
		{
			RenderStats st = job.RenderStats;                            // plain field reads, as RenderStatsPanel does while the job runs
			System.Console.Error.WriteLine($"pass {st.PassesRendered}/{options.Passes}  {st.PixelsRendered / Math.Max(clock.Elapsed.TotalSeconds, 1e-9) / 1e6:F1} Msamples/s  rays {st.RayCount}  sky {st.SkyRays}  limit {st.BounceLimitExceeded}");
		}
		RenderStats done = job.RenderStats;
		System.Console.Error.WriteLine($"{scene.Name}: {done.PassesRendered} passes of {options.RenderWidth}x{options.RenderHeight} in {clock.Elapsed.TotalSeconds:F2} s" + (cancel.IsCancellationRequested ? " (cancelled)" : ""));
		foreach ((GraphicsErrorType type, var perObject) in GraphicsValidator.Errors)      // the validator's warn-only counts (RenderJob.cs:588-607), as the GUIs list them
			System.Console.Error.WriteLine($"validator: {type} x {perObject.Values.Aggregate(0ul, (a, b) => a + b)}");

		string path = Path.GetFullPath(opt.GetValueOrDefault("out", "./image.png"));
		using (FileStream file = new(path, FileMode.Create, FileAccess.Write, FileShare.Read)) job.Image.Save(file, new PngEncoder());      // MainForm.cs:124-129
		System.Console.WriteLine(path);
		return 0;
	}

	private static int Usage(string? error)
	{
		if (error is not null) System.Console.Error.WriteLine("error: " + error);
		System.Console.Error.WriteLine("usage: --list | [--scene NAME] [--width W] [--height H] [--passes N] [--depth D] [--light-samples S] [--kmin K] [--kmax K]\n" +
		                               "       [--vis None|Normals|...] [--backend hip|cpu] [--devices 0,1,...] [--exact-lights] [--exact-occlusion] [--validate] [--out image.png]");
		return error is null ? 0 : 2;
	}
}
