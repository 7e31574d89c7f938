#!/usr/bin/env python3
"""Build-time guard of the contract the LDS-hot-set gathers of k_td_play rest on (g2048.hip, G2048_HOT_B / G2048_HOT_FENCE).

Those gathers are `global_load_dword` instructions issued from inline asm into registers the compiler believes are already
valid; one `s_waitcnt vmcnt(0)` (the fence) stands before their first use.  That is only correct while the compiler neither
spills nor copies nor reuses such a register between the load and the fence.  __graft_entry__.build() runs this script on
the shipped lib2048_hip.so and FAILS the build if, in any k_td_play<*, *, HOT = true, *>:

  1. the kernel has scratch (private_segment_fixed_size > 0) or spilled VGPRs — a spill is how an in-flight value gets saved;
  2. any instruction other than the asm forms (ds_read_b32 into, global_load_dword into, v_add_f32 from) names a VGPR at or
     above the register allocator's cap (amdgpu_waves_per_eu: v168 for the play kernels, v128 for k_eval_select_lds3);
  3. between a global_load_dword and the first s_waitcnt that guarantees its completion (vmcnt retires in order: the load is
     done once vmcnt <= the number of vector-memory instructions issued after it), any instruction reads or writes the
     load's destination register.  The rule holds for compiler-scheduled loads as well (the compiler obeys it by
     construction), so every load of the kernel is checked, not only the asm ones.

It also prints registers / scratch / LDS of every kernel with scratch, for the record.

    python tools/check_codeobj.py [path/to/lib2048_hip.so]
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = '/opt/rocm/lib/llvm/bin'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TARGET = 'hipv4-amdgcn-amd-amdhsa--gfx950'


def extract_code_objects(so, workdir):
    """Every gfx950 code object of the library: one per translation unit (the .hip_fatbin section holds their offload bundles
    back to back)."""
    fat = os.path.join(workdir, 'fatbin')
    subprocess.check_call([f'{LLVM}/llvm-objcopy', '--dump-section', f'.hip_fatbin={fat}', so, os.path.join(workdir, 'copy.so')])
    data = open(fat, 'rb').read()
    magic = b'__CLANG_OFFLOAD_BUNDLE__'
    starts = [i for i in range(len(data)) if data.startswith(magic, i)]
    out = []
    for k, a in enumerate(starts):
        part, co = os.path.join(workdir, f'bundle{k}'), os.path.join(workdir, f'code{k}.co')
        with open(part, 'wb') as f:
            f.write(data[a:starts[k + 1] if k + 1 < len(starts) else len(data)])
        subprocess.check_call([f'{LLVM}/clang-offload-bundler', '--unbundle', '--type=o', f'--input={part}', f'--targets={TARGET}', f'--output={co}'])
        out.append(co)
    return out


def extract_code_object(so, workdir, containing='k_td_play'):
    """The code object that holds the kernels named `containing` (tools/isa_mix.py and friends look at one kernel family)."""
    for co in extract_code_objects(so, workdir):
        if any(containing in k for k in kernel_notes(co)):
            return co
    raise SystemExit(f'no code object with {containing} kernels in {so}')


def kernel_notes(co):
    """{symbol: {vgpr_count, private_segment_fixed_size, vgpr_spill_count, sgpr_spill_count, group_segment_fixed_size}}"""
    text = subprocess.check_output([f'{LLVM}/llvm-readelf', '--notes', co], text=True)
    out, cur = {}, None
    for block in re.split(r'\n  - \.agpr_count:', text):
        name = re.search(r'\.name:\s+(\S+)', block)
        if not name:
            continue
        cur = {}
        for key in ('vgpr_count', 'private_segment_fixed_size', 'vgpr_spill_count', 'sgpr_spill_count', 'group_segment_fixed_size'):
            m = re.search(r'\.' + key + r':\s+(\d+)', block)
            cur[key] = int(m.group(1)) if m else -1
        out[name.group(1)] = cur
    return out


def disassemble(co, symbol):
    """[(address, text, branch target address or None)] of one kernel."""
    text = subprocess.check_output([f'{LLVM}/llvm-objdump', '-d', f'--disassemble-symbols={symbol}', co], text=True)
    ins, base = [], None
    for line in text.splitlines():
        m = re.match(r'^([0-9a-f]+) <', line)
        if m:
            base = int(m.group(1), 16)
            continue
        m = re.match(r'^\s+(\S.*?)\s*//\s*([0-9A-Fa-f]+):', line)
        if not m:
            continue
        target = None
        t = re.search(r'<[^>]*\+0x([0-9a-f]+)>\s*$', line)
        if t and m.group(1).startswith(('s_branch', 's_cbranch')):
            target = base + int(t.group(1), 16)
        ins.append((int(m.group(2), 16), m.group(1), target))
    return ins


REG = re.compile(r'\bv(\d+)\b|\bv\[(\d+):(\d+)\]')
LOADS = ('global_load', 'flat_load', 'buffer_load', 'scratch_load')
VMCNT_MAX = 63


def vregs(operands):
    regs = set()
    for m in REG.finditer(operands):
        if m.group(1) is not None:
            regs.add(int(m.group(1)))
        else:
            regs.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return regs


def is_load(op, rest):
    return op.startswith(LOADS) or ('atomic' in op and op.startswith(('global_', 'flat_', 'buffer_')) and 'sc0' in rest.split())


def check_inflight(ins):
    """For every load: walk ALL paths of the control-flow graph from the load until an s_waitcnt that guarantees its
    completion, and report every instruction on the way that names one of its destination registers.  Loads return in
    order among loads (stores and loads may pass each other, so only LATER LOADS count): the load is complete at
    `s_waitcnt vmcnt(n)` iff n <= number of loads issued after it.  Returns [(load index, offending index)]."""
    index_of = {a: i for i, (a, _, _) in enumerate(ins)}
    parsed = []
    for _, text, _ in ins:
        op, _, rest = text.partition(' ')
        parsed.append((op, rest, vregs(rest), is_load(op, rest)))
    bad = []
    for li, (op, rest, _, load) in enumerate(parsed):
        if not load:
            continue
        dest = vregs(rest.split(',')[0])
        seen, work = set(), [(li + 1, 0)]
        while work:
            pc, later = work.pop()
            while pc < len(ins):
                key = (pc, min(later, VMCNT_MAX + 1))
                if key in seen:
                    break
                seen.add(key)
                op, rest, touched, load2 = parsed[pc]
                if op.startswith('s_waitcnt'):
                    m = re.search(r'vmcnt\((\d+)\)', rest)
                    if m and int(m.group(1)) <= later:
                        break                                   # complete on this path
                    pc += 1
                    continue
                if dest & touched:
                    bad.append((li, pc))
                    break
                if load2:
                    later += 1
                if op == 's_endpgm':
                    break
                target = ins[pc][2]
                if target is not None:
                    if target in index_of:
                        work.append((index_of[target], later))
                    if op == 's_branch':
                        break
                pc += 1
    return sorted(set(bad))


def check_lds_inflight(ins, cap):
    """The other half of the asm contract: phase A reads the LDS copy into a register at or above `cap` (ds_read_b32), phase B
    lets the cold lanes overwrite it with a masked global_load_dword.  An LDS word that returns AFTER the global one would
    silently replace the table value, so on every path from such a ds_read_b32 to the first instruction that names its
    destination (the global load into it, or the v_add_f32 that consumes it) there must be an `s_waitcnt lgkmcnt(0)`.
    (lgkmcnt also counts scalar-memory loads, which return out of order: only a wait for 0 is accepted.)
    Returns [(ds_read index, offending index)]."""
    index_of = {a: i for i, (a, _, _) in enumerate(ins)}
    parsed = []
    for _, text, _ in ins:
        op, _, rest = text.partition(' ')
        parsed.append((op, rest, vregs(rest)))
    bad = []
    for li, (op, rest, _) in enumerate(parsed):
        if not op.startswith('ds_read'):
            continue
        dest = vregs(rest.split(',')[0])
        if not any(r >= cap for r in dest):
            continue
        seen, work = set(), [li + 1]
        while work:
            pc = work.pop()
            while pc < len(ins):
                if pc in seen:
                    break
                seen.add(pc)
                op, rest, touched = parsed[pc]
                if op.startswith('s_waitcnt'):
                    if re.search(r'lgkmcnt\(0\)', rest):
                        break                                   # the LDS word has landed on this path
                    pc += 1
                    continue
                if dest & touched:
                    bad.append((li, pc))
                    break
                if op == 's_endpgm':
                    break
                target = ins[pc][2]
                if target is not None:
                    if target in index_of:
                        work.append(index_of[target])
                    if op == 's_branch':
                        break
                pc += 1
    return sorted(set(bad))


def check_store_release(ins):
    """mirror_stats (k_apply_orbits*): plain stores to pinned host memory, then a workgroup barrier, then the done-counter /
    sequence word.  A workgroup-scope fence emits no vmcnt wait on gfx950, so the source asks for one explicitly; this makes
    sure it is in the code object: walking back from every s_barrier, an `s_waitcnt vmcnt(0)` must come before any
    global_store (in straight-line code; the search stops at a branch target boundary it cannot follow).
    Returns the number of s_barrier instructions that have a global_store in front of them without such a wait."""
    bad = 0
    for i, (_, text, _) in enumerate(ins):
        if not text.startswith('s_barrier'):
            continue
        j = i - 1
        while j >= 0:
            t = ins[j][1]
            if t.startswith('s_waitcnt') and re.search(r'vmcnt\(0\)', t):
                break
            if t.startswith(('global_store', 'flat_store')):
                bad += 1
                break
            if t.startswith(('s_branch', 's_cbranch', 's_endpgm', 's_barrier')):
                break
            j -= 1
    return bad


def main():
    so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, '2048_amd', 'lib2048_hip.so')
    failures = []
    with tempfile.TemporaryDirectory() as tmp:
        cos = extract_code_objects(so, tmp)
        notes, where = {}, {}
        for co_ in cos:
            for k_, v_ in kernel_notes(co_).items():
                notes[k_], where[k_] = v_, co_
        print(f'[codeobj] {len(cos)} code objects (translation units), {len(notes)} kernels')
        # the kernels whose gathers land in registers outside the allocator's range (choose_hot, choose_small<3>)
        guarded = {'k_td_play_hot': 4 * 17, 'k_td_play_lds3': 2 * 52, 'k_eval_select_lds3': 2 * 52}
        cap = {'k_td_play_hot': 168, 'k_td_play_lds3': 168, 'k_eval_select_lds3': 128}      # first VGPR outside the allocator's range (amdgpu_waves_per_eu)
        hot = [k for k in notes if any(g in k for g in guarded)]
        for g in guarded:
            if not any(g in k for k in hot):
                failures.append(f'no {g} kernel found in the code object (did the mangling change?)')
        for k in sorted(notes):
            d = notes[k]
            if d['private_segment_fixed_size'] > 0 or d['vgpr_spill_count'] > 0:
                short = re.sub(r'^_ZN12_GLOBAL__N_1\d+', '', k)[:60]
                print(f'[codeobj] note: {short}: scratch {d["private_segment_fixed_size"]} B, {d["vgpr_count"]} VGPRs, {d["vgpr_spill_count"]} spilled')
        for k in hot:
            d = notes[k]
            kind = next(g for g in guarded if g in k)
            tag = re.search(kind + r'I?((?:L[ib]\d+E)*)', k)
            params = ', '.join(re.findall(r'L[ib](\d+)E', tag.group(1))) if tag else ''
            name = f'{kind}<{params}>' if params else kind
            if d['private_segment_fixed_size'] != 0 or d['vgpr_spill_count'] != 0:
                failures.append(f'{name}: scratch {d["private_segment_fixed_size"]} B, {d["vgpr_spill_count"]} spilled VGPRs — the register '
                                f'allocator ran out below the cap, and a spilled value is how an in-flight register gets saved')
            ins = disassemble(where[k], k)
            loads = sum(1 for _, x, _ in ins if x.startswith('global_load_dword '))
            fences = sum(1 for _, x, _ in ins if x.startswith('s_waitcnt vmcnt(0)'))
            bad = check_inflight(ins)
            bad_lds = check_lds_inflight(ins, cap[kind])
            lds_reads = sum(1 for _, x, _ in ins if x.startswith('ds_read_b32 ') and any(r >= cap[kind] for r in vregs(x.split(',')[0])))
            print(f'[codeobj] {name}: {d["vgpr_count"]} VGPRs, scratch {d["private_segment_fixed_size"]} B, LDS {d["group_segment_fixed_size"]} B, '
                  f'{len(ins)} instructions, {loads} global_load_dword, {fences} full vmcnt fences, {len(bad)} in-flight register touches, '
                  f'{lds_reads} ds_read_b32 above the cap, {len(bad_lds)} touched before lgkmcnt(0)')
            if lds_reads < guarded[kind]:
                failures.append(f'{name}: only {lds_reads} ds_read_b32 into registers >= v{cap[kind]} (expected >= {guarded[kind]})')
            for li, oi in bad_lds[:5]:
                failures.append(f'{name}: "{ins[oi][1]}" names the destination of "{ins[li][1]}" on a path without s_waitcnt lgkmcnt(0) in between')
            # the registers above the cap belong to the asm statements: LDS read into, masked global load into, v_add_f32 from
            foreign = []
            for _, text, _ in ins:
                op, _, rest = text.partition(' ')
                high = [r for r in vregs(rest) if r >= cap[kind]]
                if not high:
                    continue
                ops = [o.strip() for o in rest.split(',')]
                ok = (op in ('ds_read_b32', 'global_load_dword') and vregs(ops[0]) == set(high) and not any(r >= cap[kind] for r in vregs(','.join(ops[1:])))) or \
                     (op in ('v_add_f32', 'v_add_f32_e32') and len(ops) == 3 and vregs(ops[2]) == set(high) and not any(r >= cap[kind] for r in vregs(ops[0] + ',' + ops[1])))
                if not ok:
                    foreign.append(text)
            if foreign:
                failures.append(f'{name}: {len(foreign)} instruction(s) outside the asm forms name a VGPR >= v{cap[kind]}, e.g. "{foreign[0]}" — the register allocator was not held below the cap')
            if loads < guarded[kind]:
                failures.append(f'{name}: only {loads} global_load_dword (expected >= {guarded[kind]}): is the LDS + masked-load path still there?')
            for li, oi in bad[:5]:
                failures.append(f'{name}: "{ins[oi][1]}" names the destination of "{ins[li][1]}" (issued {oi - li} instructions earlier) on a path without a covering s_waitcnt')
        for k in sorted(notes):
            if 'k_apply_orbits' not in k:
                continue
            ins = disassemble(where[k], k)
            unguarded = check_store_release(ins)
            short = 'k_apply_orbits_mean' if 'k_apply_orbits_mean' in k else 'k_apply_orbits'
            print(f'[codeobj] {short}: {sum(1 for _, x, _ in ins if x.startswith("s_barrier"))} s_barrier, {unguarded} with a store in front and no vmcnt(0) wait')
            if unguarded:
                failures.append(f'{short}: a global store reaches an s_barrier without s_waitcnt vmcnt(0) (mirror_stats publishes host-visible data behind that barrier)')
        # k_td_update_owner<N, PLAIN> (n >= 4): scalar registers spilled to VGPR lanes.  The record loops of these kernels are bound by
        # instruction issue; round 4 once shipped a form with 7 543 v_readlane_b32 in one instance (575 inside the loop a trained
        # agent's busiest workgroups run) and lost 5 % of that agent's step to it (profiles/r04_experiments.txt item 16).
        for k in sorted(notes):
            m = re.search(r'k_td_update_ownerILi([456])ELb([01])E', k)
            if not m:
                continue
            ins = disassemble(where[k], k)
            spills = sum(1 for _, x, _ in ins if x.startswith('v_readlane_b32'))
            limit = 120 if m.group(2) == '1' else 500
            print(f'[codeobj] k_td_update_owner<{m.group(1)}, {"plain" if m.group(2) == "1" else "hot-first"}>: {len(ins)} instructions, {spills} v_readlane_b32')
            if spills > limit:
                failures.append(f'k_td_update_owner<{m.group(1)}, PLAIN={m.group(2)}>: {spills} v_readlane_b32 (limit {limit}): the scalar registers no longer fit the record loops')
    if failures:
        print('\n'.join('[codeobj] FAIL: ' + f for f in failures))
        return 1
    print('[codeobj] ok: no scratch in the guarded kernels, no instruction names the destination of a load before a wait that covers it')
    return 0


if __name__ == '__main__':
    sys.exit(main())
