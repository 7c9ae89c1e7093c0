"""Small pieces every command-line tool of the package shares: input-file checks with the reference's messages
(utils_wgbs.py:355-406: what a user sees when a path is wrong) and the option groups the reference's tools have in common
(utils_wgbs.py:233-260: -s / -r / --array_id / -L / --genome, -@)."""
import os

from .genome import IllegalArgumentError


def file_problem(path, suffix=None):
    """None when `path` names an existing file (ending in `suffix`, if one is asked for), else the reference's message."""
    if path is None:
        return 'Input file is None'
    if not os.path.isfile(path):
        return f'No such file: {path}'
    if suffix is not None and not path.endswith(suffix):
        return f'file {path} must end with {suffix}'
    return None


def require_file(path, suffix=None):
    why = file_problem(path, suffix)
    if why:
        raise IllegalArgumentError(why)
    return path


def require_files(paths, min_len=1):
    """A list of existing files that all carry the first one's extension; -> that extension."""
    if len(paths) < min_len:
        raise IllegalArgumentError(f'Input error: at least {min_len} input files must be given')
    if len(paths[0]) == 1:                                     # a string where a list was meant: its "files" are single characters
        raise IllegalArgumentError(f'Input is not a list of files: {paths}')
    ext = os.path.splitext(paths[0])[1]
    for p in paths:
        require_file(p, ext)
    return ext


def lines_of(path):
    """The non-empty lines of a list file that do not start with '#', stripped."""
    require_file(path)
    with open(path) as f:
        return [ln.strip() for ln in f if ln.strip() and not ln.startswith('#')]


# (flags, keyword arguments) of the options that select what part of the genome a tool works on
_WHERE = (
    (('-s', '--sites'), dict(help='a CpG index range, of the form: "450000-450050"')),
    (('-r', '--region'), dict(help='genomic region of the form "chr1:10,000-10,500"')),
    (('--array_id',), dict(help='Illumina array id, e.g. cg00001755')),
)
_WHERE_BED = (('-L', '--bed_file'), dict(help='Bed file. Columns <chr, start, end>. '
                                              'For some features columns 4-5 should be <startCpG, endCpG> (run wgbstools convert -L BED_PATH)'))


def add_where_options(parser, required=False, bed_file=False):
    """-s | -r | --array_id (| -L), mutually exclusive, and --genome.  -> the exclusive group."""
    group = parser.add_mutually_exclusive_group(required=required)
    for flags, kw in _WHERE + ((_WHERE_BED,) if bed_file else ()):
        group.add_argument(*flags, **kw)
    parser.add_argument('--genome', help='Genome reference name. Default is "default".', default='default')
    return group


def default_threads():
    try:
        return int(os.environ['SLURM_JOB_CPUS_PER_NODE']) if 'SLURM_JOB_CPUS_PER_NODE' in os.environ else (os.cpu_count() or 8)
    except ValueError:
        return 8


def add_threads_option(parser):
    """-@ (kept for command-line compatibility: the GPU path does not fork workers; host-side thread pools read it where they exist)."""
    parser.add_argument('-@', '--threads', type=int, default=default_threads(),
                        help='Number of threads to use (default: all available CPUs)')


# what pandas.read_csv treats as missing by default (the reference reads every table with it; with comment='#' the entries that
# hold a '#' can never reach a field)
NA_TOKENS = frozenset(['', '#N/A', '#N/A N/A', '#NA', '-1.#IND', '-1.#QNAN', '-NaN', '-nan', '1.#IND', '1.#QNAN', '<NA>',
                       'N/A', 'NA', 'NULL', 'NaN', 'None', 'n/a', 'nan', 'null'])
