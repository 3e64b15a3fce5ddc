#!/usr/bin/env python3
"""Spill slots against the EXEC mask, from the gfx950 assembly of a kernel (CPU-side, no GPU needed).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S x.hip -o x.s
    tools/isa_spill_check.py x.s [--kernel REGEX] [--list] [--fail-on hazard|masked|spill]

A VGPR spill is a per-lane scratch store / load.  A store issued while EXEC is narrowed (inside an
`s_and_saveexec_b64 ... s_or_b64 exec, exec, ...` region) writes the active lanes only; a later load of that slot under a
WIDER mask reads, in the other lanes, whatever the slot held before -- a stale value from an earlier wide store at best,
uninitialised scratch (different on every run) at worst.  The compiler's own spill placement is meant to keep every such
pair consistent; this tool checks what it produced instead of trusting it:

  * builds the kernel's control-flow graph from labels and branches,
  * gives every instruction its EXEC context = the stack of saved-EXEC SGPR pairs still narrowing it
    (`s_and_saveexec_b64 sX, ...` pushes sX; `s_or_b64 exec, exec, sX` pops down to sX; `s_xor_b64 exec, exec, sX` flips an
    else-side; `s_andn2_b64 exec, exec, ...` inside loops narrows further under the loop's own save),
  * computes, per scratch slot and program point, the mask contexts under which the slot has been written on EVERY path
    to that point (forward must-analysis, join = intersection), and reports per kernel
      spills      scratch slots (dwords) the kernel uses,
      masked      scratch stores / loads issued under a narrowed EXEC,
      hazards     loads for which some path holds no store issued under a mask at least as wide as the load's own -- lanes
                  of that load may read memory no instruction of this kernel wrote.

Exit status (for the build): --fail-on spill: non-zero if a selected kernel spills at all; masked: if it spills under a narrowed
mask; hazard (default): only for loads without a covering store.
"""
import argparse
import re
import subprocess
import sys
from collections import defaultdict

LABEL = re.compile(r'^(\.LBB\d+_\d+):')
KSTART = re.compile(r'^(_Z\w+):\s')
SCR = re.compile(r'^\s+scratch_(load|store)_(dword(?:x\d)?|b\d+|ubyte|ushort|short_d16\w*)\s+(.*)$')
WIDTH = {'dword': 1, 'dwordx2': 2, 'dwordx3': 3, 'dwordx4': 4, 'b32': 1, 'b64': 2, 'b96': 3, 'b128': 4}


def demangle(names):
  out = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True).stdout.splitlines()
  return dict(zip(names, out))


def kernels(path):
  """yield (name, [lines]) for every function body in the assembly file"""
  name, body = None, []
  with open(path) as f:
    for line in f:
      m = KSTART.match(line)
      if m and name is None:
        name, body = m.group(1), []
        continue
      if name is not None:
        if line.startswith('.Lfunc_end'):
          yield name, body
          name = None
        else:
          body.append(line.rstrip('\n'))


def slot_of(args):
  """(base offset in bytes, is_sgpr_or_vgpr_addressed) of a scratch instruction's operand string"""
  off = 0
  m = re.search(r'offset:(-?\d+)', args)
  if m:
    off = int(m.group(1))
  dynamic = not re.search(r'\boff\b', args)   # `off` = no address register: a fixed frame slot
  return off, dynamic


class Ins:
  __slots__ = ('idx', 'text', 'op', 'args')

  def __init__(self, idx, text):
    self.idx, self.text = idx, text
    parts = text.split(None, 1)
    self.op = parts[0]
    self.args = parts[1] if len(parts) > 1 else ''


def analyse(body):
  # ---- basic blocks ----------------------------------------------------------------------------------------------------
  ins, label_at = [], {}
  for line in body:
    m = LABEL.match(line)
    if m:
      label_at[m.group(1)] = len(ins)
      continue
    s = line.strip()
    if not s or s.startswith(';') or s.startswith('.') or not line.startswith('\t'):
      continue
    s = s.split(';')[0].rstrip()
    if s:
      ins.append(Ins(len(ins), s))
  n = len(ins)
  leaders = {0} | set(label_at.values())
  for i, x in enumerate(ins):
    if x.op.startswith('s_cbranch') or x.op == 's_branch' or x.op == 's_endpgm':
      leaders.add(i + 1)
  leaders = sorted(l for l in leaders if l < n)
  block_of, blocks = {}, []
  for bi, l in enumerate(leaders):
    end = leaders[bi + 1] if bi + 1 < len(leaders) else n
    blocks.append((l, end))
    for i in range(l, end):
      block_of[i] = bi
  succ = defaultdict(list)
  for bi, (l, e) in enumerate(blocks):
    last = ins[e - 1]
    if last.op == 's_endpgm':
      continue
    if last.op == 's_branch':
      succ[bi].append(block_of[label_at[last.args.strip()]])
      continue
    if last.op.startswith('s_cbranch'):
      succ[bi].append(block_of[label_at[last.args.strip()]])
    if e < n:
      succ[bi].append(block_of[e])

  # ---- EXEC context per instruction: forward propagation, first state to reach a block wins (structured code agrees) -----
  def step(state, x):
    """state: tuple of 'sX@i' = saved-EXEC register pairs still narrowing EXEC (outermost first), i = the saving instruction
    (register names are reused region after region: the index makes a region's identity unique)"""
    a = x.args.replace(' ', '')

    def find(reg):
      for k in range(len(state) - 1, -1, -1):
        if state[k].split('@')[0] == reg:
          return k
      return -1

    if x.op in ('s_and_saveexec_b64', 's_andn2_saveexec_b64'):
      return state + ('%s@%d' % (a.split(',')[0], x.idx),)
    if x.op == 's_or_saveexec_b64':   # else-side entry: sX = exec; exec |= sY (widen to the region's mask), then an xor narrows
      d, s = a.split(',')[0], a.split(',')[1]
      k = find(s)
      if k >= 0:
        state = state[:k]
      return state + ('%s@%d' % (d, x.idx),)
    if x.op == 's_or_b64' and a.startswith('exec,exec,'):
      s = a.split(',')[2]
      k = find(s)
      if k < 0:
        k = find('loop:' + s)   # a divergent loop's exit: the lanes that left (accumulated in sX) come back
      return state[:k] if k >= 0 else state   # (unknown name: keep, conservative = still narrowed)
    if x.op == 's_mov_b64' and a.startswith('exec,'):
      s = a.split(',')[1]
      if s == '-1':
        return ()
      k = find(s)
      return state[:k] if k >= 0 else state
    if x.op == 's_andn2_b64' and a.startswith('exec,exec,'):   # divergent loop: lanes that are done leave, sX accumulates them
      s = 'loop:' + a.split(',')[2]
      return state if find(s) >= 0 else state + ('%s@%d' % (s, x.idx),)
    if x.op in ('s_xor_b64', 's_and_b64') and a.startswith('exec,exec,'):
      return state if state else ('?%s@%d' % (x.op, x.idx),)   # narrowing without a save we saw: mark
    return state

  state_in = {0: ()}
  state_at = [None] * n
  work = [0]
  seen = set()
  conflicts = 0
  while work:
    bi = work.pop()
    if bi in seen:
      continue
    seen.add(bi)
    st = state_in[bi]
    l, e = blocks[bi]
    for i in range(l, e):
      state_at[i] = st
      st = step(st, ins[i])
    for sb in succ[bi]:
      if sb not in state_in:
        state_in[sb] = st
        work.append(sb)
      elif state_in[sb] != st:
        # a join reached with two different contexts: keep the narrower knowledge (the common prefix = the wider mask is
        # what holds AFTER the join only if the code re-widened; structured code re-widens at the label itself)
        conflicts += 1

  # ---- scratch accesses ----------------------------------------------------------------------------------------------------
  acc = []   # (idx, kind, slot dwords tuple, state, dynamic)
  for x in ins:
    m = SCR.match('\t' + x.text)
    if not m:
      continue
    kind, wname, args = m.group(1), m.group(2), m.group(3)
    width = WIDTH.get(wname, 1)
    off, dyn = slot_of(args)
    slots = tuple(off // 4 + k for k in range(width))
    acc.append((x.idx, kind, slots, state_at[x.idx] or (), dyn))
  if not acc:
    return dict(spill_dwords=0, masked=0, hazards=[], n_acc=0, conflicts=conflicts, regions=sum(1 for x in ins if 'saveexec' in x.op), acc=[])

  # ---- per slot: the mask contexts under which it has been written ON EVERY PATH to here (must-analysis: join = intersection) --
  slots_all = sorted({s for a in acc for s in a[2]})
  by_block = defaultdict(list)
  for a in acc:
    by_block[block_of[a[0]]].append(a)
  preds = defaultdict(list)
  for b, ss in succ.items():
    for s in ss:
      preds[s].append(b)
  hazards = []

  def flow(bi, inset, report):
    cur = {s: set(v) for s, v in inset.items()}
    for a in by_block.get(bi, []):
      if a[1] == 'load':
        if report:
          for s in a[2]:
            # covered iff some store on every path ran under a context that is a prefix of (= at least as wide as) the load's
            if not any(a[3][:len(st)] == st for st in cur.get(s, ())):
              hazards.append((a[0], s, a[3], sorted(cur.get(s, ()))))
      else:
        for s in a[2]:
          if a[4]:
            continue              # register-addressed scratch (a local array, not a spill slot): not tracked
          cur.setdefault(s, set()).add(a[3])
    return cur

  OUT = {}       # block -> {slot: set(contexts)}; missing = TOP (not yet visited)
  nb = len(blocks)
  changed = True
  INS = {}
  while changed:
    changed = False
    for bi in range(nb):
      if bi == 0:
        inset = {}
      else:
        ps = [OUT[p] for p in preds[bi] if p in OUT]
        if not ps:
          continue
        inset = {}
        for s in set.intersection(*[set(p.keys()) for p in ps]):
          v = set.intersection(*[p[s] for p in ps])
          if v:
            inset[s] = v
      INS[bi] = inset
      out = flow(bi, inset, False)
      if OUT.get(bi) != out:
        OUT[bi] = out
        changed = True
  for bi in range(nb):
    if bi in INS:
      flow(bi, INS[bi], True)
  masked = sum(1 for a in acc if a[3])
  return dict(spill_dwords=len(slots_all), masked=masked, hazards=hazards, n_acc=len(acc), conflicts=conflicts,
              regions=sum(1 for x in ins if 'saveexec' in x.op), acc=acc, ins=ins)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('asm', nargs='+')
  ap.add_argument('--kernel', default='.', help='regex on the demangled kernel name')
  ap.add_argument('--list', action='store_true', help='print every masked access and hazard')
  ap.add_argument('--fail-on', choices=['hazard', 'masked', 'spill', 'never'], default='hazard')
  ap.add_argument('--quiet', action='store_true', help='print only the kernels that fail the chosen criterion')
  args = ap.parse_args()
  bad = 0
  for path in args.asm:
    ks = list(kernels(path))
    names = demangle([k for k, _ in ks])
    for name, body in ks:
      dn = names.get(name, name)
      if not re.search(args.kernel, dn):
        continue
      r = analyse(body)
      fails = (args.fail_on == 'spill' and r['spill_dwords']) or (args.fail_on == 'masked' and r['masked']) or (args.fail_on == 'hazard' and r['hazards'])
      if fails or not args.quiet:
        print('%s%-110s slots %4d  accesses %4d  masked %4d  hazards %3d  exec-regions %4d' % ('FAIL ' if fails else '', dn[:110], r['spill_dwords'], r['n_acc'], r['masked'], len(r['hazards']), r['regions']))
      if args.list:
        for a in r['acc']:
          if a[3]:
            print('    masked %-5s slots %s  exec %s  | %s' % (a[1], a[2], '/'.join(a[3]), r['ins'][a[0]].text))
      if args.list or fails:
        for h in r['hazards']:
          print('    HAZARD load @%d slot %d under exec %s; written on every path only under: %s  | %s' % (h[0], h[1], '/'.join(h[2]) or 'full', ['/'.join(c) or 'full' for c in h[3]] or 'NOTHING', r['ins'][h[0]].text))
      bad += 1 if fails else 0
  if not bad and args.quiet:
    print('spill check (%s): ok' % args.fail_on)
  return 1 if bad else 0


if __name__ == '__main__':
  sys.exit(main())
