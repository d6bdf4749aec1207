#!/usr/bin/env python3
"""Mine the reference's two scenes for fixtures: a small reader of Unity's scene YAML that extracts
what RayTraceMaster / RayTraceObject see at run time — the camera (pose, field of view), `numBounces` / `numRays`
(Assets/Scripts/RayTraceMaster.cs:17-18 as serialised in the scene) and every GameObject carrying an ENABLED RayTraceObject
component: world transform, sphere-or-mesh decision (a SphereCollider makes it an analytic sphere, RayTraceObject.cs:28-34, with
radius = collider radius x the largest lossy scale), the Unity built-in mesh it references by fileID, and its material fields
(RayTraceObject.cs:12-15).  Output: tests/golden/scene_<name>.json — DATA only (numbers and names), no reference text.

    python tests/golden/make_scene_fixtures.py [/root/reference]

Unity's built-in meshes themselves are not in the reference's tree (fileIDs 10202 cube, 10206 cylinder, 10207 sphere,
10208 capsule, 10209 plane, 10210 quad of the editor's default resources): unityraytracer_amd.scenes synthesises stand-ins.
The sky (`SkyboxTexture`, an .hdr whose blob is missing from the tree) is replaced by the procedural sky of scenes.make_sky."""
import json
import os
import re
import sys

BUILTIN_MESHES = {10202: "cube", 10206: "cylinder", 10207: "sphere", 10208: "capsule", 10209: "plane", 10210: "quad"}


def parse_documents(text):
    """Unity scene YAML -> {fileID: (class id, {key: value})}; values are scalars, flow mappings {x: 1, ...} or nested blocks
    (kept as dicts one level deep; lists of `- component: {fileID: n}` as lists)."""
    docs = {}
    cur, stack = None, None
    for line in text.splitlines():
        m = re.match(r"--- !u!(\d+) &(\d+)", line)
        if m:
            cur = {}
            docs[int(m.group(2))] = (int(m.group(1)), cur)
            continue
        if cur is None or not line.strip() or line.startswith("%"):
            continue
        indent = len(line) - len(line.lstrip(" "))
        body = line.strip()
        if indent == 0:                                   # "GameObject:" — the document's type line
            continue
        if body.startswith("- "):                         # list item under the last key at lower indent
            item = parse_value(body[2:].split(": ", 1)[1]) if ": " in body[2:] else parse_value(body[2:])
            key = cur.get("__last_list__")
            if key is not None:
                cur[key].append(item)
            continue
        if ": " in body or body.endswith(":"):
            k, _, v = body.partition(":")
            v = v.strip()
            if indent == 2:
                if v == "":
                    cur[k] = []
                    cur["__last_list__"] = k
                else:
                    cur[k] = parse_value(v)
    for _, d in docs.values():
        d.pop("__last_list__", None)
    return docs


def parse_value(v):
    v = v.strip()
    if v.startswith("{") and v.endswith("}"):
        out = {}
        for part in re.findall(r"(\w+):\s*([^,}]+)", v):
            out[part[0]] = parse_value(part[1])
        return out
    try:
        return int(v)
    except ValueError:
        pass
    try:
        return float(v)
    except ValueError:
        return v


def quat_mul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return (aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz)


def quat_rotate(q, v):
    x, y, z, w = q
    vx, vy, vz = v
    tx, ty, tz = 2 * (y * vz - z * vy), 2 * (z * vx - x * vz), 2 * (x * vy - y * vx)
    return (vx + w * tx + (y * tz - z * ty), vy + w * ty + (z * tx - x * tz), vz + w * tz + (x * ty - y * tx))


def world_transform(docs, tid):
    """position, rotation (x, y, z, w), lossy scale of Transform `tid` (parents composed; the scenes are flat, so this is the
    local transform in practice; non-uniform parent scale under rotation is not handled and does not occur)."""
    _, t = docs[tid]
    p = t["m_LocalPosition"]; r = t["m_LocalRotation"]; s = t["m_LocalScale"]
    pos, rot, scale = (p["x"], p["y"], p["z"]), (r["x"], r["y"], r["z"], r["w"]), (s["x"], s["y"], s["z"])
    father = t.get("m_Father", {}).get("fileID", 0) if isinstance(t.get("m_Father"), dict) else 0
    if father:
        fp, fr, fs = world_transform(docs, father)
        scaled = (pos[0] * fs[0], pos[1] * fs[1], pos[2] * fs[2])
        rp = quat_rotate(fr, scaled)
        pos = (fp[0] + rp[0], fp[1] + rp[1], fp[2] + rp[2])
        rot = quat_mul(fr, rot)
        scale = (scale[0] * fs[0], scale[1] * fs[1], scale[2] * fs[2])
    return pos, rot, scale


def mine(path, rto_guid=None):
    docs = parse_documents(open(path).read())
    by_go = {}                                            # GameObject id -> {class id: [component dicts]}
    for fid, (cls, d) in docs.items():
        go = d.get("m_GameObject")
        if isinstance(go, dict) and go.get("fileID"):
            by_go.setdefault(go["fileID"], {}).setdefault(cls, []).append(d)
    out = {"source": os.path.basename(path), "objects": []}
    for go_id, comps in by_go.items():
        cls, go = docs[go_id]
        name, active = go.get("m_Name"), go.get("m_IsActive", 1)
        tr = [fid for fid, (c, d) in docs.items() if c == 4 and isinstance(d.get("m_GameObject"), dict) and d["m_GameObject"].get("fileID") == go_id]
        pos, rot, scale = world_transform(docs, tr[0]) if tr else ((0, 0, 0), (0, 0, 0, 1), (1, 1, 1))
        for mb in comps.get(114, []):
            if "numBounces" in mb:                        # the RayTraceMaster on the camera
                out["numBounces"], out["numRays"] = mb["numBounces"], mb["numRays"]
            script = mb.get("m_Script", {}).get("guid") if isinstance(mb.get("m_Script"), dict) else None
            if "albedoColor" in mb or (script is not None and script == rto_guid):     # a RayTraceObject
                # SampleScene was saved by an older RayTraceObject without the colour fields: Unity then keeps the script's field
                # initialisers (RayTraceObject.cs:12-15)
                defaults = {"albedoColor": {"r": 0.0, "g": 0.4, "b": 1.0}, "specularColor": {"r": 0.7, "g": 0.0, "b": 1.0},
                            "emissionColor": {"r": 0.0, "g": 0.0, "b": 0.0}}
                col = lambda c: [mb.get(c, defaults[c])["r"], mb.get(c, defaults[c])["g"], mb.get(c, defaults[c])["b"]]
                sphere = [c for c in comps.get(135, [])]
                mesh = comps.get(33, [{}])[0].get("m_Mesh", {})
                obj = {"name": name, "enabled": bool(mb.get("m_Enabled", 1)) and bool(active),
                       "type": "sphere" if sphere else "mesh",
                       "mesh": BUILTIN_MESHES.get(mesh.get("fileID"), str(mesh.get("fileID"))),
                       "position": list(pos), "rotation": list(rot), "scale": list(scale),
                       "albedoColor": col("albedoColor"), "specularColor": col("specularColor"), "emissionColor": col("emissionColor"),
                       "smoothness": mb.get("smoothness", 0.69), "serialized_material": "albedoColor" in mb}
                if sphere:                                # RayTraceObject.cs:33: radius = collider.radius * max(lossyScale)
                    obj["collider_radius"] = sphere[0]["m_Radius"]
                    obj["radius"] = sphere[0]["m_Radius"] * max(scale)
                out["objects"].append(obj)
        for cam in comps.get(20, []):
            out["camera"] = {"position": list(pos), "rotation": list(rot), "field_of_view": cam.get("field of view"),
                             "near": cam.get("near clip plane"), "far": cam.get("far clip plane")}
    out["objects"].sort(key=lambda o: o["name"])
    return out


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    here = os.path.dirname(os.path.abspath(__file__))
    meta = open(os.path.join(ref, "Assets", "Scripts", "RayTraceObject.cs.meta")).read()
    rto_guid = re.search(r"guid:\s*(\w+)", meta).group(1)     # which MonoBehaviours are RayTraceObjects
    for scene in ("Scene1", "SampleScene"):
        data = mine(os.path.join(ref, "Assets", "Scenes", scene + ".unity"), rto_guid)
        dst = os.path.join(here, f"scene_{scene}.json")
        json.dump(data, open(dst, "w"), indent=1, sort_keys=True)
        en = [o for o in data["objects"] if o["enabled"]]
        print(f"{scene}: camera {data.get('camera')}, numBounces {data.get('numBounces')} numRays {data.get('numRays')}, "
              f"{sum(o['type'] == 'sphere' for o in en)} spheres + {sum(o['type'] == 'mesh' for o in en)} meshes enabled "
              f"({len(data['objects']) - len(en)} disabled) -> {dst}")


def screenshot_stats(ref, here):
    """The reference's only visual evidence: Screenshots/<Time.time>-<sample>.png.  A few NUMBERS per capture —
    size, mean colour of the bottom and the top quarter, the row where the brown ground plane (RS:167) starts — for the qualitative
    checks of tests/test_reference_scenes.py; the 8-bit images themselves stay in the reference."""
    from PIL import Image
    import numpy as np
    out = {}
    for name in ("25.64697-62.png", "282.4567-2866.png", "14.58841-1320.png"):
        path = os.path.join(ref, "Screenshots", name)
        if not os.path.exists(path):
            continue
        a = np.asarray(Image.open(path).convert("RGB"), dtype=np.float64) / 255.0
        h, w = a.shape[:2]
        rows = a.mean(axis=1)                              # mean colour per row, top to bottom
        lum = rows @ np.array([0.2126, 0.7152, 0.0722])
        k = max(2, h // 100)                               # the horizon: the sharpest drop of row brightness (sky above, ground below)
        drop = np.array([lum[y - k:y].mean() - lum[y:y + k].mean() for y in range(k, h - k)])
        below = [int(np.argmax(drop)) + k]
        out[name] = {"width": w, "height": h, "samples": int(name.split("-")[1].split(".")[0]),
                     "ground_mean_rgb": [round(float(x), 4) for x in a[int(0.75 * h):].mean(axis=(0, 1))],
                     "sky_mean_rgb": [round(float(x), 4) for x in a[:int(0.25 * h)].mean(axis=(0, 1))],
                     "ground_starts_at_row_fraction": round(below[0] / h, 4) if below else None}
    json.dump(out, open(os.path.join(here, "screenshot_stats.json"), "w"), indent=1, sort_keys=True)
    print("screenshots:", out)


if __name__ == "__main__":
    main()
    screenshot_stats(sys.argv[1] if len(sys.argv) > 1 else "/root/reference", os.path.dirname(os.path.abspath(__file__)))
