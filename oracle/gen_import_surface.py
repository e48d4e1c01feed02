"""Test infrastructure (build container only; reads /root/reference): record which names the reference's Python files import from
the five un-vendored third-party packages this repo replaces (flash_mla_fp8, flash_mla_swap, deep_gemm, flashinfer, eps) ->
tests/golden/import_surface.json.  tests/test_import_surface_cpu.py then asserts that every (module, name) pair either resolves in
the shim under sglang-fluentllm_amd/ or is listed in INTEGRATION.md as deliberately absent.

A record = {module, name, file, line, level}: `from M import N` gives (M, N); `import M.sub [as x]` gives (M.sub, None);
attribute chains on an imported module object (`M.a.b` after `import M` / `from P import M`) give (M, "a.b").  level = "module"
when the statement executes at import time of the file (incl. inside try / if at module level), "function" when it is deferred into
a def.  HOT = the files SURVEY.md section 8 (a)/(b) cite (the path); everything else under python/sglang is recorded with
hot = false for information."""
import ast
import json
import os
import sys

REF = "/root/reference/python/sglang"
PKGS = ("flash_mla_fp8", "flash_mla_swap", "deep_gemm", "flashinfer", "eps")
HOT = """srt/layers/attention/flashmla_backend.py srt/layers/attention/flashinfer_mla_backend.py srt/layers/attention/utils.py
srt/mem_cache/memory_pool.py srt/mem_cache/allocator.py srt/layers/moe/gemms/fp8/fire.py srt/layers/moe/executors/fp8_eps_executor.py
srt/layers/moe/executors/eps_executor.py srt/layers/moe/executors/deep_ep_executor.py srt/tbo/tbo_executor.py
srt/layers/moe/dispatcher/fast_ep.py srt/layers/moe/topk.py srt/layers/moe/layer.py
srt/layers/moe/layouts/fp8.py srt/layers/dense/gemms/fp8/deep_geem.py srt/layers/dense/gemms/fp8/fp8_kernel.py
srt/layers/dense/gemms/fp8/fp8_utils.py srt/layers/dense/layouts/fp8.py srt/layers/flashinfer_comm_fusion.py
srt/layers/dp_attention.py srt/layers/layernorm.py srt/layers/activation.py srt/distributed/parallel_state.py
srt/distributed/decoder_comm_manager.py srt/_custom_ops.py srt/models/deepseek_v2.py
srt/models/utils.py""".split()


def top(mod):
    return mod.split(".")[0]


def scan(path, rel):
    try:
        tree = ast.parse(open(path, encoding="utf-8").read())
    except SyntaxError:
        return []
    out, aliases = [], {}     # alias name -> module path it stands for

    def visit(node, level):
        for ch in ast.iter_child_nodes(node):
            lv = "function" if isinstance(ch, (ast.FunctionDef, ast.AsyncFunctionDef, ast.Lambda)) else level
            if isinstance(ch, ast.Import):
                for a in ch.names:
                    if top(a.name) in PKGS:
                        out.append(dict(module=a.name, name=None, file=rel, line=ch.lineno, level=level))
                        aliases[a.asname or top(a.name)] = a.name if a.asname else top(a.name)
            elif isinstance(ch, ast.ImportFrom) and ch.module and ch.level == 0 and top(ch.module) in PKGS:
                for a in ch.names:
                    out.append(dict(module=ch.module, name=a.name, file=rel, line=ch.lineno, level=level))
                    aliases[a.asname or a.name] = ch.module + "." + a.name
            visit(ch, lv)

    visit(tree, "module")
    # attribute chains rooted at an imported module object
    for node in ast.walk(tree):
        if isinstance(node, ast.Attribute):
            chain, cur = [], node
            while isinstance(cur, ast.Attribute):
                chain.append(cur.attr)
                cur = cur.value
            if isinstance(cur, ast.Name) and cur.id in aliases and top(aliases[cur.id]) in PKGS:
                out.append(dict(module=aliases[cur.id], name=".".join(reversed(chain)), file=rel, line=node.lineno, level="attribute"))
    return out


def main():
    recs = []
    for root, _, files in os.walk(REF):
        for f in sorted(files):
            if f.endswith(".py"):
                p = os.path.join(root, f)
                rel = os.path.relpath(p, REF)
                for r in scan(p, rel):
                    r["hot"] = rel in HOT
                    recs.append(r)
    # longest attribute chain per (file, line, module) only: a.b.c also walks a.b
    keep, seen = [], set()
    for r in sorted(recs, key=lambda r: (r["file"], r["line"], r["module"], -(len(r["name"] or "")))):
        if r["level"] == "attribute":
            k = (r["file"], r["line"], r["module"])
            if any(s[:3] == k and (s[3] or "").startswith(r["name"]) for s in seen):
                continue
            seen.add(k + (r["name"],))
        keep.append(r)
    missing_hot = [f for f in HOT if not os.path.exists(os.path.join(REF, f))]
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "import_surface.json")
    json.dump(dict(packages=PKGS, hot_files=HOT, hot_files_missing=missing_hot, records=keep), open(out, "w"), indent=0)
    print(f"{len(keep)} records ({sum(r['hot'] for r in keep)} on hot-path files) -> {out}; hot files not found: {missing_hot}")


if __name__ == "__main__":
    sys.exit(main())
