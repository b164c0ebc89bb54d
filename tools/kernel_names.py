"""Kernel-name helpers shared by traffic_parse.py and step_traffic.py: rocprofv3 prints some template instantiations demangled and some mangled."""
import re


def instantiation(name):
    """The 3x3 halo kernel runs as several template instantiations with different work per launch (fp16 forward, bf16 data gradient, the
    folded skip convolution, the sub-pixel phase forms): each gets its OWN record, keyed `conv3x3_halo_ws_kernel<f16>`, `<bf16>`, `<f16,+skip>` ...
    (rocprofv3 prints some instantiations demangled - bf16 as "bool _Accum, bool, E" - and some mangled)."""
    base = short(name)
    if base == "conv_wgrad_slots_ws_kernel":      # <LOOK, kXF16, kShare, kS2>: the stride-2 four-plane form (round 6) forms 4 - 16 MFMAs per step
        m = re.search(r"kernelI((?:L[bi]\d+E)+)E", name) if name.startswith("_Z") else re.search(r"kernel<(.*?)>\(", name)      # where the others form 36
        lits = re.findall(r"L[bi](\d+)E", m.group(1)) if m and name.startswith("_Z") else [t.strip() for t in m.group(1).split(",")] if m else []
        return f"{base}<stride-2 planes>" if len(lits) >= 4 and lits[3] in ("1", "true") else None
    if not base.startswith("conv3x3_halo") and not base.startswith("conv_subpixel"):
        return None
    if name.startswith("_Z"):
        m = re.search(r"kernelI(DF16_|DF16b)((?:L[bi]n?\d+E)*)E", name)
        if not m:
            return None
        typ = "f16" if m.group(1) == "DF16_" else "bf16"
        args = re.findall(r"L([bi])(n?)(\d+)E", m.group(2))
        vals = [-int(v) if neg else int(v) for _, neg, v in args]
    else:
        m = re.search(r"kernel<(.*)>\(", name)
        if not m:
            return None
        body = m.group(1)
        typ = "bf16" if "_Accum" in body or "bfloat" in body or "__bf16" in body else "f16"
        vals = [1 if t.strip() == "true" else 0 if t.strip() == "false" else int(t) for t in body.split(",") if t.strip() in ("true", "false") or t.strip().lstrip("-").isdigit()]
        if "_Accum" in body and base == "conv3x3_halo_ws_kernel":
            vals = [1] + vals      # the garbled bf16 form ("bool _Accum, bool, E, 16, ...") has lost the first literal, kPrefetchW (true in every shipped launch)
    tags = [typ]
    if base == "conv3x3_halo_ws_kernel":          # <T, kPrefetchW, kShape, kFuse, kSkip, kStamp, kMerge, kRes> (the merged / unmerged folded forms and the
                                                  # residual / no-residual plain forms share a key: same work per launch)
        if len(vals) >= 3 and vals[2]:
            tags.append("+gn")
        if len(vals) >= 4 and vals[3]:
            tags.append("+skip")
    elif base == "conv_subpixel_ws_kernel":       # <T, kMode>
        # rocprofv3's demangler garbles the FIRST literal behind a bf16 type argument ("<bool _Accum, int, E>": the value of Li1E is lost, while
        # Li2E stays mangled and is parsed above).  The counter passes therefore run with --mangled-kernels since round 6 (tools/profile_r06.sh);
        # a demangled bf16 name without a value stays "?" instead of being guessed (it was taken for mode 1, wrong for bf16 activations)
        tags.append({0: "upsample", 1: "transposed", 2: "upsample dgrad"}.get(vals[0] if vals else -1, "?"))
    else:
        tags += [str(v) for v in vals]
    return f"{base}<{','.join(tags)}>"


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"_ZN12_GLOBAL__N_1(\d+)", name)
    if m:
        n = int(m.group(1)); start = m.end()
        name = name[start:start + n]
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0].split("<")[0].strip()
