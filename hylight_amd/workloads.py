"""The synthetic workloads of BASELINE.json `configs` (recipes: SURVEY.md section 8d), shared by bench.py, the
full-size GPU tests and the tools.  Seeded (20241008); nothing here is timed.

    C2  10 000 reads, mean 8 kb, 5 strains x 400 kb (1 % SNPs + 0.05 % indels), --nsplit 100          configs[1]
    C3  100 000 reads, mean 10 kb, 20 strains x 1 Mb, ANI 98.5-99.5 % per strain, --nsplit 200         configs[2]
    C4  1 M long reads + 10 M paired 2x250 short reads, 100 strains x 2 Mb, --nsplit 1000              configs[3]
    C5  500 000 reads, 50 strains x 2 Mb, ANI 95-99 %, 1 / 0.5 / 0.5 % read errors, iden 0.90 len 1500  configs[4]

`scale` shrinks reads AND genome length together (same depth, same divergence, same error model): the variants the
tests run when the full configuration does not fit their time budget.
"""
from __future__ import annotations

import os

from . import simulate as S

STAGE_MAIN = dict(len_over=6000, mc=2, iden=0.95)          # script/HyLight.py:130 (len_over is hard-coded there)

CONFIGS = {
    "C2": dict(sim=dict(n_strains=5, genome_len=400_000, n_reads=10_000, mean_len=8_000, snp_rate=0.01,
                        strain_indel_rate=0.0005, err_sub=0.003, err_ins=0.001, err_del=0.001),
               nsplit=100, stage=STAGE_MAIN),
    "C3": dict(sim=dict(n_strains=20, genome_len=1_000_000, n_reads=100_000, mean_len=10_000, snp_rate=(0.005, 0.015),
                        strain_indel_rate=0.0005, err_sub=0.003, err_ins=0.001, err_del=0.001),
               nsplit=200, stage=STAGE_MAIN),
    "C4": dict(sim=dict(n_strains=100, genome_len=2_000_000, n_reads=1_000_000, mean_len=10_000, snp_rate=0.01,
                        strain_indel_rate=0.0005, err_sub=0.003, err_ins=0.001, err_del=0.001),
               short=dict(n_pairs=5_000_000, read_len=250, insert_mean=450.0, insert_sd=27.0, err_sub=0.001),
               nsplit=1000, stage=STAGE_MAIN, stage_short=dict(len_over=70, mc=3, iden=0.95)),       # HyLight.py:200,207
    # --min_identity 0.90 --min_ovlp_len 1500: the constants the later calls of the run take (HyLight.py:149,168,180);
    # the main call's len_over stays 6000 whatever --min_ovlp_len says - both are exercised
    "C5": dict(sim=dict(n_strains=50, genome_len=2_000_000, n_reads=500_000, mean_len=10_000, snp_rate=(0.01, 0.05),
                        strain_indel_rate=0.0005, err_sub=0.01, err_ins=0.005, err_del=0.005),
               nsplit=60, stage=dict(len_over=1500, mc=2, iden=0.90)),
}


PARALLEL_FROM = 200_000          # read sets from this size on are written by simulate_reads_to_fasta (per-block seeds)


def c1_reads():
    """C1 (BASELINE.json configs[0]; the example files it names are not in the reference tree): C2's
    recipe at a tenth of its size - 1 000 long reads on 5 strains x 40 kb.  Run as `--corrected --nsplit 100 -t 8`."""
    reads, _ = S.simulate_reads(seed=S.SEED_DEFAULT, min_len=1_000, max_len=40_000, **config("C2", 0.1)["sim"])
    return reads


def config(name, scale=1.0):
    """The recipe `name`, optionally shrunk: reads and genome length times `scale` (depth unchanged)."""
    c = CONFIGS[name]
    sim = dict(c["sim"])
    if scale != 1.0:
        sim["n_reads"] = max(1, int(round(sim["n_reads"] * scale)))
        sim["genome_len"] = max(50_000, int(round(sim["genome_len"] * scale)))
    out = dict(c, sim=sim, name=name if scale == 1.0 else f"{name}@{scale:g}")
    if "short" in c and scale != 1.0:
        out["short"] = dict(c["short"], n_pairs=max(1, int(round(c["short"]["n_pairs"] * scale))))
    return out


def describe(cfg):
    s = cfg["sim"]
    ani = s["snp_rate"]
    ani = f"SNP {ani[0]:g}-{ani[1]:g} per strain" if isinstance(ani, tuple) else f"SNP {ani:g}"
    txt = (f"{cfg['name']}: {s['n_reads']} synthetic ONT reads, mean {s['mean_len']} bp, {s['n_strains']} strains x "
           f"{s['genome_len']} bp ({ani}, indel {s['strain_indel_rate']:g}), errors {s['err_sub']:g}/{s['err_ins']:g}/"
           f"{s['err_del']:g}, ava, --nsplit {cfg['nsplit']}")
    if "short" in cfg:
        txt += f" + {2 * cfg['short']['n_pairs']} short reads (2x{cfg['short']['read_len']})"
    return txt


def make_long(cfg, path, seed=S.SEED_DEFAULT):
    """Writes the long reads as 2-line FASTA (what filter_non_atcg leaves, utils.py:81-114); returns
    (n_reads, n_bases, strains)."""
    if cfg["sim"]["n_reads"] >= PARALLEL_FROM:      # the full C4 / C5: drawn by a pool of processes, never held in memory
        n, bases, strains = S.simulate_reads_to_fasta(path + ".tmp", seed=seed, min_len=1_000, max_len=40_000, **cfg["sim"])
        os.replace(path + ".tmp", path)
        return n, bases, strains
    reads, strains = S.simulate_reads(seed=seed, min_len=1_000, max_len=40_000, **cfg["sim"])
    S.write_fasta(reads, path + ".tmp")
    os.replace(path + ".tmp", path)
    return len(reads), int(sum(len(r.seq) for r in reads)), strains


def make_short(cfg, strains, path, seed=S.SEED_DEFAULT + 1):
    if 2 * cfg["short"]["n_pairs"] >= 4 * PARALLEL_FROM:        # the full C4: 10 M reads, written by a pool of processes
        n = S.simulate_short_pairs_to_fasta(path + ".tmp", seed, strains, **cfg["short"])
        os.replace(path + ".tmp", path)
        return n
    reads = S.simulate_short_pairs(seed, strains, **cfg["short"])
    S.write_fasta(reads, path + ".tmp")
    os.replace(path + ".tmp", path)
    return len(reads)
