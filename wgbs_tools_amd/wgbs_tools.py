#!/usr/bin/python3 -u
"""`wgbstools <command>` dispatcher (reference: src/python/wgbs_tools.py:50-79).  This build carries `segment` — the
MI355X-native hot path — and the steps either side of it that work on the same resident data: `convert` (loci <-> CpG
indexes, what feeds `segment -L`), `pat2beta` (the producer of the beta files), `beta_to_blocks`, `beta_to_table` and `find_markers` (the reductions over the blocks it writes);
every other reference subcommand is out of scope and says so."""
import sys

from .genome import IllegalArgumentError, eprint

VERSION = '0.2.0-mi355x'
COMMANDS = ['segment', 'convert', 'pat2beta', 'beta_to_blocks', 'beta_to_table', 'find_markers']
# reference command list (wgbs_tools.py:11-48), for the "not in this build" message
REFERENCE_ONLY = ['view', 'merge', 'cview', 'index', 'beta_cov',
                  'beta2bed', 'beta2bw', 'bam2pat', 'mbias', 'init_genome', 'set_default_ref', 'vis', 'pat_fig',
                  'homog', 'test_bimodal', 'compare_betas', 'dmb', 'mix_pat', 'bed2beta',
                  'split_by_allele', 'split_by_meth', 'frag_len', 'add_cpg_counts', 'beta_to_450k']


def print_help():
    msg = '\nUsage: wgbstools <command> [<args>]'
    msg += '\nrun wgbstools <command> -h for more information'
    msg += '\nOptional commands:\n'
    for key in COMMANDS:
        msg += '\t' + key + '\n'
    print(msg)
    return 1


def main(argv=None):
    argv = list(sys.argv if argv is None else argv)
    if len(argv) < 2 or argv[1] in ('-h', '--help'):
        return print_help()
    if argv[1] in ('--version', '-v'):
        print('wgbstools (MI355X segment) version', VERSION)
        return 0
    cmd = argv[1]
    if cmd in REFERENCE_ONLY:
        eprint(f'wgbstools {cmd}: not part of this build (it carries the MI355X-native `segment`, `convert`, `beta_to_blocks`, `beta_to_table`)')
        return 1
    if cmd not in COMMANDS:
        eprint('Invalid command:', f'\033[01;31m{cmd}\033[00m')
        return print_help()
    try:
        import importlib
        importlib.import_module('.' + cmd, __package__).main(argv[2:])
        return 0
    except IllegalArgumentError as e:          # wgbs_tools.py:77-79
        eprint(f'Invalid input argument\n{e}')
        return 1


if __name__ == '__main__':
    sys.exit(main())
