"""The data format on the input side of the path: the package's bundled data sets (data/*.RData, three
small gzip'd RDX3 files, committed here as data fixtures) read without R, against the text fixtures the
round-1 decoder wrote from the same files and the facts SURVEY.md section 8c records about them."""
import os

import numpy as np
import pytest

from bmm_mcmc_amd import rdata
from util import GOLDEN, load_dataset


@pytest.mark.parametrize("name,shape,total", [("K2_N100_P5", (100, 5), 212), ("K2_N1000_P5", (1000, 5), 2128),
                                               ("K3_N1000_P5", (1000, 5), 2172)])
def test_bundled_datasets_decode_to_the_committed_matrices(name, shape, total):
    objs = rdata.read_rdata(os.path.join(GOLDEN, name + ".RData"))
    assert list(objs) == [name]                       # one object, tagged with the data set's name
    X = rdata.read_rdata_matrix(os.path.join(GOLDEN, name + ".RData"))
    assert X.shape == shape and X.dtype == np.int32 and X.flags.f_contiguous
    assert int(X.sum()) == total and set(np.unique(X)) <= {0, 1}
    assert np.array_equal(X, load_dataset(name))


def test_first_rows_are_the_ones_the_survey_recorded():
    X = rdata.read_rdata_matrix(os.path.join(GOLDEN, "K2_N100_P5.RData"))
    assert X[:4].tolist() == [[1, 1, 0, 0, 0], [0, 1, 0, 0, 0], [1, 1, 0, 0, 0], [0, 0, 0, 0, 0]]
    assert X[99].tolist() == [0, 0, 1, 1, 0]
    np.testing.assert_allclose(X.mean(axis=0), [.54, .59, .36, .33, .30])


def test_other_files_are_refused(tmp_path):
    p = tmp_path / "x.RData"
    import gzip
    with gzip.open(p, "wb") as f:
        f.write(b"RDA2\nA\n")
    with pytest.raises(ValueError, match="RDX3"):
        rdata.read_rdata(str(p))
