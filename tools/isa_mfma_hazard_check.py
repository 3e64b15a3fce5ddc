#!/usr/bin/env python3
"""MFMA result hazards along EVERY control-flow path, from the gfx950 assembly of a build (CPU-side, no GPU needed).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S x.hip -o x.s
    tools/isa_mfma_hazard_check.py x.s [--kernel REGEX] [--list]          (exit status 1 if any kernel has a finding)

Why this exists.  gfx950 does not interlock a matrix instruction's result registers: between a
`v_mfma_*` and the first instruction that READS its destination registers (a store's data, an LDS write's data, a vector-ALU
source, another MFMA's A / B / overlapping C operand) software must leave a number of wait states that grows with the MFMA's
pass count; the compiler's hazard recognizer inserts `s_nop`s for that.  Its backward search over predecessor blocks marks a
block as visited the first time it reaches it -- whatever distance it has accumulated on that path -- so at a join whose two
incoming paths both pass through the block holding the MFMA it prices the LONGER path only.  The build of
`gcn_chain_t_bwd_kernel<192,4,false>` that returned a different dA on every run had exactly that:

        v_mfma_f32_16x16x4_f32 v[16:19], v39, v43, v[16:19]      ; dacc[0][2] += ..., 8 passes
        s_cbranch_vccnz .LBB51_1405                              ; jb = 3 skipped: the document has three row blocks
        ...                                                      ; (the fall-through path: 4 more MFMAs = 8 wait states)
    .LBB51_1405:
        s_or_b64 exec, exec, s[82:83]
        s_nop 1                                                  ; enough for the fall-through path only
        scratch_store_dwordx4 off, v[16:19], off offset:32       ; spill of dacc[0][2]: 4 wait states after the MFMA, 10 needed

-- the spill store read registers 2 and 3 of the accumulator before the MFMA's last passes had written them.

This tool redoes the search without that shortcut: for every VGPR an instruction reads it walks backwards along all paths
(bounded by the largest requirement, so it is cheap), counts wait states the way the hardware does (one per instruction,
N + 1 for `s_nop N`), and reports every MFMA write that is closer than required.

Required wait states (gfx950, matching what the compiler enforces on straight-line code in this repository's builds; f32
MFMAs are "SGEMM" class, low-precision ones "XDL" class and need one more):
    consumer = VALU / VMEM / LDS / export read, or MFMA SrcA / SrcB:   passes + 2   (XDL: passes + 3)
    consumer = MFMA SrcC overlapping the result but not identical:     passes       (XDL: passes + 1)
    consumer = MFMA SrcC identical to the result tuple:                0 (back-to-back accumulation is interlocked)
"""
import argparse
import re
import subprocess
import sys
from collections import defaultdict

LABEL = re.compile(r'^(\.LBB\d+_\d+):')
KSTART = re.compile(r'^(_Z\w+):\s')
VREG = re.compile(r'\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]')   # vector and accumulation registers (a-registers: number + 1000)

PASSES = [
    (re.compile(r'v_mfma_f32_32x32x2_?f32'), 16), (re.compile(r'v_mfma_f32_32x32x1_?(2b_)?f32'), 16),
    (re.compile(r'v_mfma_f32_16x16x4_?f32'), 8), (re.compile(r'v_mfma_f32_16x16x1_?(4b_)?f32'), 8),
    (re.compile(r'v_mfma_f32_4x4x1_?(16b_)?f32'), 2),
]


def mfma_passes(op):
  for rx, p in PASSES:
    if rx.match(op):
      return p, False
  # anything else (f16 / bf16 / i8 / fp8 / xf32 / f64): priced as a 16-pass XDL instruction -- conservative
  return 16, True


def regs(tok):
  out = []
  for m in VREG.finditer(tok):
    if m.group(1) is not None:
      out.append(int(m.group(2)) + (1000 if m.group(1) == 'a' else 0))
    else:
      base = 1000 if m.group(3) == 'a' else 0
      out.extend(range(base + int(m.group(4)), base + int(m.group(5)) + 1))
  return out


def split_operands(args):
  out, depth, cur = [], 0, ''
  for ch in args:
    if ch == '[':
      depth += 1
    elif ch == ']':
      depth -= 1
    if ch == ',' and depth == 0:
      out.append(cur.strip())
      cur = ''
    else:
      cur += ch
  if cur.strip():
    out.append(cur.strip())
  return out


class Ins:
  __slots__ = ('idx', 'text', 'op', 'ops', 'ws', 'writes', 'reads', 'mfma', 'srcc', 'line')

  def __init__(self, idx, text, line):
    self.idx, self.text, self.line = idx, text, line
    parts = text.split(None, 1)
    self.op = parts[0]
    self.ops = split_operands(parts[1]) if len(parts) > 1 else []
    self.ws = 1
    if self.op == 's_nop':
      self.ws = int(self.ops[0], 0) + 1
    self.mfma = self.op.startswith('v_mfma') or self.op.startswith('v_smfmac')
    self.writes, self.reads, self.srcc = [], [], []
    op = self.op
    if not self.ops:
      return
    stores = op.startswith(('global_store', 'scratch_store', 'flat_store', 'buffer_store', 'ds_write', 'ds_store', 'exp', 'global_atomic', 'flat_atomic', 'buffer_atomic'))
    no_vdst = stores or op.startswith(('s_', 'v_cmp', 'v_cmpx', 'ds_gws', 'buffer_wbl2', 'buffer_inv', 'global_wb', 'global_inv')) or op in ('v_nop', 'v_readfirstlane_b32', 'v_readlane_b32')
    if op in ('v_readfirstlane_b32', 'v_readlane_b32'):
      for o in self.ops[1:]:
        self.reads += regs(o)
      return
    if op.startswith('v_cmp') and not op.startswith('v_cmpx') and self.ops and not self.ops[0].startswith('v'):
      for o in self.ops[1:]:
        self.reads += regs(o)
      return
    if no_vdst:
      for o in self.ops:
        self.reads += regs(o)
      return
    # first operand = destination (a second destination for carry-outs is an SGPR pair / vcc: no VGPRs)
    self.writes = regs(self.ops[0])
    if self.mfma:
      self.reads = regs(self.ops[1]) + regs(self.ops[2])
      self.srcc = regs(self.ops[3]) if len(self.ops) > 3 else []
    else:
      for o in self.ops[1:]:
        self.reads += regs(o)
      # instructions that read their destination as well
      if op.startswith(('v_fmac', 'v_mac', 'v_dot', 'v_pk_fmac')) or '_dpp' in op or 'sdwa' in op or op.startswith(('v_writelane', 'v_cndmask')) and False:
        self.reads += self.writes


def kernels(path):
  name, body, start = None, [], 0
  with open(path) as f:
    for ln, line in enumerate(f, 1):
      m = KSTART.match(line)
      if m and name is None:
        name, body, start = m.group(1), [], ln
        continue
      if name is not None:
        if line.startswith('.Lfunc_end'):
          yield name, body, start
          name = None
        else:
          body.append(line.rstrip('\n'))


def analyse(body, first_line):
  ins, label_at = [], {}
  for k, line in enumerate(body):
    m = LABEL.match(line)
    if m:
      label_at[m.group(1)] = len(ins)
      continue
    if not line.startswith('\t'):
      continue
    s = line.strip()
    if not s or s.startswith(';') or s.startswith('.'):
      continue
    s = s.split(';')[0].rstrip()
    if s:
      ins.append(Ins(len(ins), s, first_line + 1 + k))
  n = len(ins)
  # predecessors at instruction granularity: fall-through + branch edges
  preds = defaultdict(list)
  for i, x in enumerate(ins):
    if x.op == 's_endpgm':
      continue
    if x.op == 's_branch':
      preds[label_at[x.ops[0]]].append(i)
      continue
    if x.op.startswith('s_cbranch'):
      preds[label_at[x.ops[0]]].append(i)
    if i + 1 < n:
      preds[i + 1].append(i)

  findings = []
  LIMIT = 20
  for x in ins:
    wanted = []     # (register, kind) kind: 'ab' (full requirement) or 'c'
    for r in set(x.reads):
      wanted.append((r, 'read'))
    for r in set(x.srcc):
      wanted.append((r, 'srcc'))
    if not wanted:
      continue
    for r, kind in wanted:
      # depth-first over (instruction, accumulated wait states); a state is pruned only if the same instruction was already
      # reached with FEWER OR EQUAL accumulated wait states (the thing the compiler's search gets wrong)
      best = {}
      stack = [(p, 0) for p in preds[x.idx]]
      while stack:
        i, wsacc = stack.pop()
        if wsacc >= LIMIT:
          continue
        if i in best and best[i] <= wsacc:
          continue
        best[i] = wsacc
        y = ins[i]
        if r in y.writes:
          if y.mfma:
            passes, xdl = mfma_passes(y.op)
            if kind == 'srcc':
              if x.mfma and sorted(x.srcc) == sorted(y.writes):
                need = 0
              else:
                need = passes + (1 if xdl else 0)
            else:
              need = passes + (3 if xdl else 2)
            if wsacc < need:
              findings.append((x, y, r, wsacc, need))
          continue      # (any other writer: the MFMA before it is no longer what this read sees)
        stack.extend((p, wsacc + y.ws) for p in preds[i])
  # one finding per (consumer, producer)
  seen, out = set(), []
  for f in findings:
    k = (f[0].idx, f[1].idx)
    if k in seen:
      continue
    seen.add(k)
    out.append(f)
  return out, sum(1 for x in ins if x.mfma)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('asm', nargs='+')
  ap.add_argument('--kernel', default='.', help='regex on the demangled kernel name')
  ap.add_argument('--list', action='store_true')
  args = ap.parse_args()
  bad = 0
  for path in args.asm:
    ks = list(kernels(path))
    names = [k for k, _, _ in ks]
    dem = dict(zip(names, subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True).stdout.splitlines()))
    for name, body, start in ks:
      dn = dem.get(name, name)
      if not re.search(args.kernel, dn):
        continue
      f, nm = analyse(body, start)
      if f or args.list:
        print('%s: %-100s mfma %5d  hazards %d' % (path.split('/')[-1], dn[:100], nm, len(f)))
      for x, y, r, have, need in f:
        print('    %s%d: line %d `%s`  <- %d wait states (need %d) <-  line %d `%s`' % ('a' if r >= 1000 else 'v', r % 1000, y.line, y.text, have, need, x.line, x.text))
      bad += 1 if f else 0
  if not bad:
    print('no MFMA result is read too early on any path')
  return 1 if bad else 0


if __name__ == '__main__':
  sys.exit(main())
