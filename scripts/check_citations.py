#!/usr/bin/env python3
"""Checks every reference citation `Name.cs:a-b` (or `:a`, or `:a,b,c`) in oracle/, include/, raytracer_amd/ and csharp/ against the
mounted reference tree (build container only: /root/reference does not exist on the GPU box).

For each citation: the file must exist (matched by base name, or by the cited path suffix when the name is ambiguous), the lines
must lie inside it, and the cited lines (+- SLACK) must mention something the citing line talks about — an identifier of the citing
line (CamelCase words; snake_case names are also tried as CamelCase) or, when the citing line names nothing checkable, any code at all
(not blank / not only a brace).  Prints one line per problem and exits 1 if there is any.

usage: python scripts/check_citations.py [-v]"""
import os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
SLACK = 3
SCAN = ["oracle/oracle.cpp", "include/rt_hip.h", "raytracer_amd/csrc/rt_device.h", "raytracer_amd/csrc/rt_kernels.h", "raytracer_amd/csrc/rt_lib.hip",
        "raytracer_amd/csrc/rt_bvh.h", "raytracer_amd/csrc/rt_bvh.cpp", "raytracer_amd/csrc/rt_lbvh.hip", "csharp/HipRenderBackend.cs",
        "csharp/Impl/DeviceFlatteners.cs", "csharp/Impl/DistanceAttenuation.cs", "csharp/Impl/Triangle.cs", "csharp/Console/Program.cs", "raytracer_amd/render_job.py",
        "raytracer_amd/core.py", "raytracer_amd/impl.py", "raytracer_amd/builtin_scenes.py", "raytracer_amd/flatten.py"]
CITE = re.compile(r"((?:[A-Za-z_.]+/)*[A-Za-z][A-Za-z0-9_]*\.cs):(\d+(?:-\d+)?(?:,\s?\d+(?:-\d+)?)*)")
STOP = {"RenderJob", "Core", "Impl", "Hittable", "Hittables", "Material", "Materials", "Lights", "Textures", "Vector3", "Colour", "HitRecord", "Ray", "Scene",
        "SURVEY", "DESIGN", "NOT", "NEW", "TryHit", "The", "This"}


def index_reference():
    by_name = {}
    for d, _, files in os.walk(REF):
        for f in files:
            if f.endswith(".cs"):
                by_name.setdefault(f, []).append(os.path.join(d, f))
    return by_name


def camel(s):
    return "".join(p.capitalize() for p in s.split("_"))


def identifiers(line, cite_span):
    text = line[:cite_span[0]] + " " + line[cite_span[1]:]
    words = set(re.findall(r"[A-Za-z_][A-Za-z0-9_]{3,}", text))
    out = set()
    for w in words:
        if w.endswith(".cs") or w in STOP:
            continue
        if re.search(r"[a-z][A-Z]", w) or (w[0].isupper() and len(w) >= 5):
            out.add(w)
        elif "_" in w and w.islower():
            out.add(camel(w))
    return out


def main():
    verbose = "-v" in sys.argv
    if not os.path.isdir(REF):
        print("reference tree not mounted: nothing checked"); return 0
    by_name = index_reference()
    problems, checked = [], 0
    for rel in SCAN:
        path = os.path.join(ROOT, rel)
        if not os.path.exists(path):
            continue
        for ln, line in enumerate(open(path, errors="replace").read().splitlines(), 1):
            for m in CITE.finditer(line):
                cited, ranges = m.group(1), m.group(2)
                name = os.path.basename(cited)
                cands = by_name.get(name, [])
                if len(cands) > 1 and "/" in cited:
                    tail = cited.replace("Core/", "Ararem.RayTracer.Core/").replace("Impl/", "Ararem.RayTracer.Impl/").replace("Dev/", "Ararem.RayTracer.Display.Dev/").replace("Eto/", "Ararem.RayTracer.Display.EtoForms/")
                    cands = [c for c in cands if c.endswith(tail)] or cands
                if not cands:
                    if name in ("Triangle.cs", "HipRenderBackend.cs", "DeviceFlatteners.cs", "DistanceAttenuation.cs"):
                        continue                       # files this repo adds to the reference
                    problems.append(f"{rel}:{ln}: {cited} is not in the reference tree"); continue
                src = open(cands[0], errors="replace").read().splitlines()
                ids = identifiers(line, m.span())
                for r in re.split(r",\s?", ranges):
                    a, _, b = r.partition("-")
                    a = int(a); b = int(b) if b else a
                    checked += 1
                    if a < 1 or b > len(src) or b < a:
                        problems.append(f"{rel}:{ln}: {cited}:{r} is outside the file ({len(src)} lines)"); continue
                    window = "\n".join(src[max(0, a - 1 - SLACK):min(len(src), b + SLACK)])
                    if not re.search(r"[A-Za-z]", window):
                        problems.append(f"{rel}:{ln}: {cited}:{r} cites blank lines"); continue
                    whole = "\n".join(src)
                    cls = {os.path.splitext(os.path.basename(c.group(1)))[0] for c in CITE.finditer(line)}      # class names of the files cited on this line say nothing
                    here = lambda i, t: re.search(r"\b" + re.escape(i) + r"\b", t)
                    # what the citation is FOR: the clause in front of it (back to the previous citation or sentence break).  Checked when it
                    # names a method / property as the reference spells it (>= 2 humps, >= 10 characters, present in the cited file)
                    before = line[:m.start()]
                    prev = [c.end() for c in CITE.finditer(before)]
                    clause = re.split(r";| — |\. |: ", before[prev[-1] if prev else 0:])[-1]
                    strong = [i for i in identifiers(clause + " ", (len(clause), len(clause))) - cls if len(i) >= 10 and len(re.findall(r"[A-Z]", i)) >= 2 and here(i, whole)]
                    if strong and not any(here(i, window) for i in strong):
                        problems.append(f"{rel}:{ln}: {cited}:{r} — {sorted(strong)[:3]} occur in the file, but not in the cited lines")
                    elif verbose:
                        print(f"ok {rel}:{ln}: {cited}:{r} {sorted(strong)[:3]}")
    for p in problems:
        print(p)
    print(f"{checked} citations checked, {len(problems)} problem(s)")
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main())
